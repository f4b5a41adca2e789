"""Session batcher on the CPU kernel simulator: slot allocation, per-row reset, stream masks, routing."""
import numpy as np
import pytest

from tests import batcher_cases


def test_batcher_matches_the_schedule_driven_by_hand(sim_lib):
    batcher_cases.check_batcher_matches_manual_api("cpu", sim_lib)


def test_session_is_independent_of_its_neighbours(sim_lib):
    batcher_cases.check_session_independent_of_neighbours("cpu", sim_lib)


def test_slots_and_buffer_limits(sim_lib):
    batcher_cases.check_slots_and_buffers("cpu", sim_lib)


def test_guided_sessions_through_the_batcher(sim_lib):
    batcher_cases.check_batcher_with_guidance("cpu", sim_lib)


def test_batched_asr_text_only(sim_lib):
    batcher_cases.check_asr_batcher("cpu", sim_lib)


def test_io_threads_push_and_pop_while_the_loop_steps(sim_lib):
    """open / push / pop / close from I/O threads while one thread runs the model loop (the deployment shape: websocket
    handlers + batched_asr.rs's model_loop thread): nothing is lost, every channel gets its frames in order."""
    import threading
    import time

    import numpy as np
    from moshi_amd.batcher import SessionBatcher
    slots, n_frames = 3, 5
    mimi, lm, mcfg, lcfg = batcher_cases.tiny_pair("cpu", sim_lib, slots)
    F = mcfg.frame_size
    results, errors = {}, []
    with SessionBatcher(mimi, lm, slots, use_sampling=False) as b:
        stop = threading.Event()

        def model_loop():
            try:
                while not stop.is_set():
                    if b.step() == 0:
                        time.sleep(0.001)
            except Exception as e:      # noqa: BLE001
                errors.append(e)

        def client(i):
            try:
                rng = np.random.default_rng(i)
                ch = b.open()
                got = []
                for f in range(n_frames):
                    x = (0.3 * rng.standard_normal(F)).astype(np.float32)
                    for piece in np.array_split(x, 3):          # audio trickles in
                        b.push(ch, piece)
                        time.sleep(0.002)
                deadline = time.time() + 60
                while len(got) < n_frames - lcfg.max_delay and time.time() < deadline:
                    fr = b.pop(ch)
                    if fr is None:
                        time.sleep(0.005)
                    else:
                        got.append(fr)
                b.close(ch)
                results[i] = got
            except Exception as e:      # noqa: BLE001
                errors.append(e)
        loop = threading.Thread(target=model_loop)
        loop.start()
        clients = [threading.Thread(target=client, args=(i,)) for i in range(slots)]
        for t in clients:
            t.start()
        for t in clients:
            t.join()
        stop.set()
        loop.join()
        st = b.stats()
    assert not errors, errors
    assert st["frames"] == slots * n_frames and st["used_slots"] == 0
    for i in range(slots):
        assert len(results[i]) == n_frames - lcfg.max_delay
        assert all(np.isfinite(p).all() and (t >= 0).all() for p, t in results[i])


def test_a_vanished_channel_has_its_own_status(sim_lib):
    """push / pop / close on a channel id that names no open channel return MMI_ERR_NO_CHANNEL, which the binding maps
    to `UnknownChannel` (a ValueError) - no matching on the message text; a REAL argument error stays a plain ValueError."""
    from moshi_amd.batcher import SessionBatcher
    from moshi_amd.errors import UnknownChannel
    mimi, lm, mcfg, _ = batcher_cases.tiny_pair("cpu", sim_lib, 2)
    b = SessionBatcher(mimi, lm, slots=2)
    ch = b.open()
    b.close(ch)
    for call in (lambda: b.pop(ch), lambda: b.push(ch, np.zeros(mcfg.frame_size, np.float32)), lambda: b.close(ch)):
        with pytest.raises(UnknownChannel):
            call()
    assert issubclass(UnknownChannel, ValueError)
