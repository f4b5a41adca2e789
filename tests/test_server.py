"""The websocket shell around the session batcher (moshi_amd/server.py): wire format of
rust/protocol.md, the PCM codec seam, and concurrent fake clients against the engine on the CPU kernel simulator."""
import asyncio

import numpy as np
import pytest

from moshi_amd import server as srv
from moshi_amd.errors import UnknownChannel
from tests import batcher_cases


def test_wire_format_round_trips():
    assert srv.decode_message(srv.encode_handshake(7)) == (srv.MT_HANDSHAKE, (0, 7))
    assert srv.decode_message(b"\x00") == (srv.MT_HANDSHAKE, (0, 0))            # the reference's bare handshake byte
    assert srv.decode_message(srv.encode_audio(b"abc")) == (srv.MT_AUDIO, b"abc")
    assert srv.decode_message(srv.encode_text(" héllo")) == (srv.MT_TEXT, " héllo")
    assert srv.decode_message(srv.encode_control(srv.CONTROL_RESTART)) == (srv.MT_CONTROL, 3)
    assert srv.decode_message(srv.encode_metadata('{"a": 1}')) == (srv.MT_METADATA, '{"a": 1}')
    assert srv.decode_message(srv.encode_error("boom")) == (srv.MT_ERROR, "boom")
    assert srv.decode_message(bytes([srv.MT_PING])) == (srv.MT_PING, None)
    assert srv.decode_message(b"") is None and srv.decode_message(b"\x09xyz") is None   # unknown kinds are discarded
    assert srv.encode_handshake(1)[:1] == b"\x00" and len(srv.encode_handshake(1)) == 9  # u8 + 2 x u32 little endian


def test_pcm_codec_reassembles_samples_split_across_messages():
    codec = srv.PcmCodec()
    x = np.random.default_rng(0).standard_normal(1000).astype(np.float32)
    raw = codec.writer().append_pcm(x)
    r = codec.reader()
    got = np.concatenate([r.append_bytes(raw[:1001]), r.append_bytes(raw[1001:1003]), r.append_bytes(raw[1003:])])
    assert np.array_equal(got, x)


def test_opus_codec_fails_loudly_without_sphn():
    try:
        import sphn  # noqa: F401
        pytest.skip("sphn is installed here")
    except ImportError:
        pass
    with pytest.raises(RuntimeError, match="sphn"):
        srv.OpusCodec()


async def _client(session, url, frames, piece, n_expect, timeout=120.0):
    """One fake client: handshake, then its PCM trickling in as unaligned pieces; collects audio + text until n_expect frames."""
    got_pcm, got_text, hello = [], [], None
    async with session.ws_connect(url) as ws:
        msg = await ws.receive(timeout=timeout)
        hello = srv.decode_message(msg.data)
        if hello[0] == srv.MT_ERROR:
            return hello, got_pcm, got_text
        raw = np.concatenate(frames).astype("<f4").tobytes()
        for at in range(0, len(raw), piece):
            await ws.send_bytes(srv.encode_audio(raw[at:at + piece]))
        await ws.send_bytes(b"\x09ignored")                      # unknown message kind: must be discarded
        reader = srv.PcmCodec().reader()
        while len(got_pcm) < n_expect:
            msg = await ws.receive(timeout=timeout)
            kind, payload = srv.decode_message(msg.data)
            if kind == srv.MT_AUDIO:
                got_pcm.append(reader.append_bytes(payload))
            elif kind == srv.MT_TEXT:
                got_text.append(payload)
            else:
                raise AssertionError(f"unexpected message {kind} {payload!r}")
        try:                                                     # the last frame's text piece follows its audio
            while True:
                msg = await ws.receive(timeout=1.0)
                dec = srv.decode_message(msg.data) if isinstance(msg.data, bytes) else None
                if dec is None or dec[0] != srv.MT_TEXT:
                    break
                got_text.append(dec[1])
        except asyncio.TimeoutError:
            pass
    return hello, got_pcm, got_text


def test_concurrent_clients_each_get_their_own_stream(sim_lib):
    """Three websocket clients at once on a three-slot server, a fourth is turned away with an Error message; every client's
    audio is bit-identical to the same PCM pushed through a SessionBatcher alone (the server adds framing, nothing else), and
    the text pieces are the text tokens of its stream."""
    import aiohttp
    from aiohttp import web
    from moshi_amd.batcher import SessionBatcher
    slots, n_frames = 3, 5
    mimi, lm, mcfg, lcfg = batcher_cases.tiny_pair("cpu", sim_lib, slots)
    F = mcfg.frame_size
    rng = np.random.default_rng(3)
    inputs = [[(0.3 * rng.standard_normal(F)).astype(np.float32) for _ in range(n_frames)] for _ in range(slots)]

    # what each session must produce: the same frames through the batcher, one session at a time
    expect = []
    for i in range(slots):
        with SessionBatcher(mimi, lm, slots, use_sampling=False) as b:
            ch = b.open()
            b.push(ch, np.concatenate(inputs[i]))
            outs = []
            while b.step() > 0:
                while True:
                    fr = b.pop(ch)
                    if fr is None:
                        break
                    outs.append(fr)
            expect.append(outs)
    n_out = len(expect[0])
    assert 0 < n_out < n_frames                                   # first frame skipped (server.py:127-134), one step of ring delay

    async def scenario():
        with SessionBatcher(mimi, lm, slots, use_sampling=False) as batcher:
            server = srv.BatchedServer(batcher, model_version=3)
            server.start()
            runner = web.AppRunner(server.make_app())
            await runner.setup()
            site = web.TCPSite(runner, "127.0.0.1", 0)
            await site.start()
            port = site._server.sockets[0].getsockname()[1]
            url = f"http://127.0.0.1:{port}/api/chat"
            try:
                async with aiohttp.ClientSession() as session:
                    tasks = [asyncio.ensure_future(_client(session, url, inputs[i], 1000 + 37 * i, n_out)) for i in range(slots)]
                    await asyncio.sleep(0.3)                      # all three slots are taken now
                    late = await _client(session, url, inputs[0], 4096, 0)
                    results = await asyncio.gather(*tasks)
            finally:
                server.stop()
                await runner.cleanup()
            assert not server.errors, server.errors
            return results, late
    results, late = asyncio.run(scenario())
    assert late[0][0] == srv.MT_ERROR and "no free slot" in late[0][1]
    for i, (hello, pcm, text) in enumerate(results):
        assert hello == (srv.MT_HANDSHAKE, (0, 3))
        assert len(pcm) == n_out
        for f in range(n_out):
            assert np.array_equal(pcm[f], expect[i][f][0]), f"client {i} frame {f}: audio differs from the batcher's own output"
        want_text = [str(int(tok[0])) for _, tok in expect[i] if int(tok[0]) not in (0, 3)]
        assert text == want_text


def test_a_vanished_channel_does_not_stop_the_model_loop():
    """A handler may close (disconnect) or re-open (RESTART) a channel between the loop's `closed` check and its pop - the
    batcher then reports an unknown channel.  That must end nothing but this session's drain: the loop keeps stepping and
    the other sessions keep receiving (server.py model loop)."""
    import threading
    import time

    class FakeBatcher:
        frame_size = 4

        def __init__(self):
            self.steps = 0
            self.queues = {2: [("pcm", np.zeros(9, np.int64))] * 3}

        def step(self):
            self.steps += 1
            return 1 if self.steps < 50 else 0

        def pop(self, channel):
            if channel == 1:
                raise UnknownChannel("unknown channel (mmi status -1)")
            q = self.queues.get(channel, [])
            return q.pop(0) if q else None

    class Loop:
        def __init__(self):
            self.calls = []

        def call_soon_threadsafe(self, fn, arg):
            self.calls.append(arg)

    server = srv.BatchedServer(FakeBatcher(), model_version=3, idle_sleep=0.001)
    gone, alive = Loop(), Loop()
    server._sessions[1] = srv._Session(1, None, gone)
    server._sessions[2] = srv._Session(2, type("Q", (), {"put_nowait": None})(), alive)
    server.start()
    deadline = time.time() + 5
    while server.batcher.steps < 50 and time.time() < deadline:
        time.sleep(0.01)
    server.stop()
    assert not server.errors, server.errors
    assert server.batcher.steps >= 50
    assert len(alive.calls) == 3 and not gone.calls


def test_any_other_invalid_argument_is_not_swallowed_by_the_model_loop():
    """Only the unknown-channel status ends a session's drain quietly; every other MMI_ERR_INVALID (a ValueError) is a real
    failure of the model loop and is recorded as one."""
    import time

    class FakeBatcher:
        frame_size = 4
        steps = 0

        def step(self):
            self.steps += 1
            return 1

        def pop(self, channel):
            raise ValueError("null argument (mmi status -1)")

    server = srv.BatchedServer(FakeBatcher(), model_version=3, idle_sleep=0.001)
    loop = type("L", (), {"call_soon_threadsafe": lambda self, fn, arg: None})()
    server._sessions[1] = srv._Session(1, type("Q", (), {"put_nowait": None})(), loop)
    server.start()
    deadline = time.time() + 5
    while not server.errors and time.time() < deadline:
        time.sleep(0.01)
    server.stop()
    assert server.errors and isinstance(server.errors[0], ValueError) and not isinstance(server.errors[0], UnknownChannel)
