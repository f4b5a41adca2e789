"""Duplex pipeline (include/moshi_mi.h mmi_duplex_*) on the CPU kernel simulator: same bits as the serial serving loop."""
import pytest

from tests import duplex_cases


@pytest.mark.parametrize("use_sampling", [False, True])
def test_pipelined_frames_equal_the_serial_loop(sim_lib, use_sampling):
    duplex_cases.check_pipeline_is_bit_identical("cpu", sim_lib, use_sampling=use_sampling)


def test_pipeline_crosses_the_attention_program_switch(sim_lib, monkeypatch):
    """The LM's change of step program (decode attention with / without its merge launch) in the middle of a pipelined run, the
    ring split forced onto the tiny model: same bits as the serial loop (the two-graph form of it: tests/test_c_duplex_gpu.py)."""
    monkeypatch.setenv("MMI_ATTN_NS", "3")
    monkeypatch.setenv("MMI_ATTN_SOLO", "5")
    duplex_cases.check_pipeline_is_bit_identical("cpu", sim_lib, use_sampling=True)


def test_decode_reads_a_column_slice_in_place(sim_lib):
    duplex_cases.check_strided_decode("cpu", sim_lib)


def test_duplex_refuses_models_that_are_not_streaming(sim_lib):
    from moshi_amd.duplex import DuplexStream
    from moshi_amd.lm import LMGen
    from tests.batcher_cases import tiny_pair
    mimi, lm, _, _ = tiny_pair("cpu", sim_lib, 2)
    with pytest.raises(RuntimeError):
        DuplexStream(mimi, LMGen(lm))


def test_device_clock_stamps_follow_the_pipeline_order(sim_lib):
    """mmi_duplex_get_stamps (diagnostics): with the timeline on, every hand-off point of the last four frames carries a stamp of
    one clock, and they come in the order the pipeline's dependencies demand - input before encode, encode before the LM's
    step, the depth-transformer phase inside the step, decode after it - while the tokens and PCM stay what they are without
    the stamping kernels."""
    import numpy as np
    import torch
    from moshi_amd.duplex import DuplexStream
    from moshi_amd.lm import LMGen
    from tests.batcher_cases import tiny_pair
    B = 2
    mimi, lm, mcfg, lcfg = tiny_pair("cpu", sim_lib, B)
    rng = np.random.default_rng(5)
    frames = [torch.from_numpy((0.1 * rng.standard_normal((B, 1, mcfg.frame_size))).astype(np.float32)) for _ in range(6)]

    def run(stamped):
        gen = LMGen(lm, use_sampling=False)
        with mimi.streaming(B), gen.streaming(B):
            dup = DuplexStream(mimi, gen)
            if stamped:
                dup.timeline(True)
            outs = [dup.step(x) for x in frames]
            dup.join()
            st = dup.stamps() if stamped else None
            return [None if o is None or o[0] is None else (o[0].numpy().copy(), o[1].numpy().copy()) for o in outs], st
    plain, _ = run(False)
    stamped, st = run(True)
    for a, b in zip(plain, stamped):
        assert (a is None) == (b is None)
        if a is not None:
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    assert sorted(st) == [2, 3, 4, 5]
    for f, s in st.items():
        for before, after in (("in", "enc0"), ("enc0", "enc1"), ("enc1", "wait1"), ("wait0", "wait1"), ("wait1", "lm0"), ("lm0", "phase"),
                              ("phase", "lm1"), ("lm1", "dec0"), ("dec0", "dec1")):
            assert s[before] < s[after], (f, before, after, s)
    assert st[4]["lm1"] < st[5]["lm0"] and st[4]["enc1"] < st[5]["enc0"]       # stream order across frames


def test_step_rejects_a_batch_that_is_not_the_streaming_batch(sim_lib):
    """`DuplexStream.step` with fewer rows than the streaming batch must fail like `lm_gen.step` does
    (AssertionError, lm.py:679-682) instead of encoding rows it was never given; the C entry refuses it on its own too."""
    import ctypes as C
    import torch
    from moshi_amd import _capi
    from moshi_amd.duplex import DuplexStream
    from moshi_amd.lm import LMGen
    from tests.batcher_cases import tiny_pair
    B = 3
    mimi, lm, mcfg, _ = tiny_pair("cpu", sim_lib, B)
    gen = LMGen(lm, use_sampling=False)
    with mimi.streaming(B), gen.streaming(B):
        dup = DuplexStream(mimi, gen)
        assert sim_lib.mmi_duplex_batch(dup._handle) == B
        with pytest.raises(AssertionError):
            dup.step(torch.zeros(B - 1, 1, mcfg.frame_size))
        x = torch.zeros(B, 1, mcfg.frame_size)
        valid = C.c_int32(0)
        rc = sim_lib.mmi_duplex_submit(dup._handle, x.data_ptr(), dup._pcm[0].data_ptr(), None, B - 1, C.byref(valid), None)
        assert rc == _capi.MMI_ERR_SHAPE
        assert dup.step(x) == (None, None)          # the refused calls left the pipeline usable
        dup.join()
        dup.close()


def test_the_callers_input_buffer_may_be_refilled_in_place(sim_lib):
    """the frame's input is copied into the pipeline's ring in stream order, so a caller that reuses ONE
    preallocated input tensor (safe with `mimi.encode`) gets the same bits as one that hands over a fresh tensor per frame."""
    import numpy as np
    import torch
    from moshi_amd.duplex import DuplexStream
    from moshi_amd.lm import LMGen
    from tests.batcher_cases import tiny_pair
    B = 2
    mimi, lm, mcfg, _ = tiny_pair("cpu", sim_lib, B)
    rng = np.random.default_rng(11)
    frames = [(0.1 * rng.standard_normal((B, 1, mcfg.frame_size))).astype(np.float32) for _ in range(6)]

    def run(reuse):
        gen = LMGen(lm, use_sampling=False)
        buf = torch.zeros(B, 1, mcfg.frame_size)
        with mimi.streaming(B), gen.streaming(B):
            dup = DuplexStream(mimi, gen)
            outs = []
            for f in frames:
                if reuse:
                    buf.copy_(torch.from_numpy(f))
                    outs.append(dup.step(buf))
                    buf.fill_(float("nan"))          # the caller's buffer is free again as soon as step() returns
                else:
                    outs.append(dup.step(torch.from_numpy(f)))
            dup.join()
            res = [None if o[0] is None else (o[0].numpy().copy(), o[1].numpy().copy()) for o in outs]
            dup.close()
            return res
    for a, b in zip(run(False), run(True)):
        assert (a is None) == (b is None)
        if a is not None:
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


def test_a_hooked_lmgen_is_refused(sim_lib):
    """LMGen's hooks work on the caller's stream while the pipeline steps the LM on its own - refused
    (NotImplementedError) rather than racing; without hooks the same objects run."""
    import torch
    from moshi_amd.duplex import DuplexStream
    from moshi_amd.lm import LMGen
    from tests.batcher_cases import tiny_pair
    B = 2
    mimi, lm, mcfg, _ = tiny_pair("cpu", sim_lib, B)
    gen = LMGen(lm, use_sampling=False, on_text_hook=lambda t: None)
    x = torch.zeros(B, 1, mcfg.frame_size)
    with mimi.streaming(B), gen.streaming(B):
        dup = DuplexStream(mimi, gen)
        with pytest.raises(NotImplementedError):
            dup.step(x)
        dup.close()


def _replica(rank, out):
    """One of N replicas of the deployment (one process per GPU, swarm-config.yml:57-63), on the simulator: the pipelined frame
    loop of tests/duplex_cases.py, with the host thread pinned like bench.py pins it."""
    import hashlib
    import os
    import sys
    from pathlib import Path
    import numpy as np
    import torch
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    os.environ["HIPSIM_WORKERS"] = "2"
    torch.set_num_threads(1)
    import bench
    from moshi_amd import _capi
    from moshi_amd.lm import LMGen
    from tests import duplex_cases
    from tests.batcher_cases import tiny_pair
    from tests.hipsim.build_sim import LIB
    bench.pin_host_thread(rank, 8)
    lib = _capi.load(LIB)
    B, steps = 2, 7
    mimi, lm, mcfg, _ = tiny_pair("cpu", lib, B)
    rng = np.random.default_rng(7)
    frames = [(0.1 * rng.standard_normal((B, 1, mcfg.frame_size))).astype(np.float32) for _ in range(steps)]
    gen = LMGen(lm, use_sampling=True, temp=0.8, temp_text=0.7, top_k=5, top_k_text=5, seed=99)
    with mimi.streaming(B), gen.streaming(B):
        res = duplex_cases._pipelined(mimi, gen, frames, {}, torch.device("cpu"), 3)
    h = hashlib.sha256()
    for r in res:
        h.update(b"none" if r is None else r[0].tobytes() + r[1].tobytes())
    out[rank] = h.hexdigest()


def test_eight_replicas_in_eight_processes_stay_bit_identical_under_contention(sim_lib):
    """the pipeline's gate is a host wait, so N ranks on one node are N host threads competing for cores.
    Eight processes, each with its own simulator-backed DuplexStream over the same seeded session group, oversubscribing this
    machine's cores: every replica must produce the very same tokens and PCM as the serial loop run alone."""
    import hashlib
    import numpy as np
    import torch
    import torch.multiprocessing as mp
    from moshi_amd.lm import LMGen
    from tests import duplex_cases
    from tests.batcher_cases import tiny_pair
    B, steps = 2, 7
    mimi, lm, mcfg, _ = tiny_pair("cpu", sim_lib, B)
    rng = np.random.default_rng(7)
    frames = [(0.1 * rng.standard_normal((B, 1, mcfg.frame_size))).astype(np.float32) for _ in range(steps)]
    gen = LMGen(lm, use_sampling=True, temp=0.8, temp_text=0.7, top_k=5, top_k_text=5, seed=99)
    with mimi.streaming(B), gen.streaming(B):
        ref = duplex_cases._serial(mimi, gen, frames, {}, torch.device("cpu"))
    h = hashlib.sha256()
    for r in ref:
        h.update(b"none" if r is None else r[0].tobytes() + r[1].tobytes())
    out = mp.Manager().dict()
    mp.spawn(_replica, args=(out,), nprocs=8, join=True)
    assert len(out) == 8 and set(out.values()) == {h.hexdigest()}, dict(out)
