"""Websocket front end for many live full-duplex sessions on one GPU.

The reference's Python server (moshi/moshi/server.py:73-169) serves ONE session under a lock; its Rust server batches.  This is
the Python server's wire behaviour - the binary protocol of rust/protocol.md:1-31, handshake first, audio in / audio + text
pieces out - in front of `SessionBatcher` (native model loop, moshi_amd/csrc/batcher.hip), so that every connection is one slot
of the batched frame step:

    websocket handler (asyncio)            model loop (one thread)                 websocket handler
    MT=1 payload -> codec.reader -> PCM -> batcher.push(ch)  ...  batcher.step()  ->  batcher.pop(ch) -> codec.writer -> MT=1
                                                                                                      -> text piece   -> MT=2

The audio codec is a seam: the reference speaks Opus in Ogg pages through `sphn` (server.py:85,90,119-122), which is not in
this image, so the shipped codec is `PcmCodec` (raw little-endian float32 samples at 24 kHz; same framing, same message type).
`OpusCodec` binds `sphn` when it can be imported and fails loudly otherwise.  Nothing here touches the GPU directly.
"""
from __future__ import annotations

import asyncio
import struct
import threading
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, Optional, Tuple

import numpy as np

from .errors import UnknownChannel

# ---- wire format (rust/protocol.md:7-31) ---------------------------------------------------------------------------------------
MT_HANDSHAKE, MT_AUDIO, MT_TEXT, MT_CONTROL, MT_METADATA, MT_ERROR, MT_PING = range(7)
CONTROL_START, CONTROL_END_TURN, CONTROL_PAUSE, CONTROL_RESTART = range(4)
PROTOCOL_VERSION = 0


def encode_handshake(model_version: int = 0) -> bytes:
    """MT=0: protocol version (u32, always 0) + model version (u32), little endian."""
    return bytes([MT_HANDSHAKE]) + struct.pack("<II", PROTOCOL_VERSION, int(model_version))


def encode_audio(payload: bytes) -> bytes:
    return bytes([MT_AUDIO]) + payload


def encode_text(text: str) -> bytes:
    return bytes([MT_TEXT]) + text.encode("utf8")


def encode_control(control: int) -> bytes:
    return bytes([MT_CONTROL, int(control)])


def encode_metadata(json_text: str) -> bytes:
    return bytes([MT_METADATA]) + json_text.encode("utf8")


def encode_error(description: str) -> bytes:
    return bytes([MT_ERROR]) + description.encode("utf8")


def decode_message(message: bytes) -> Optional[Tuple[int, object]]:
    """(message type, decoded payload), or None for an empty / unknown message ("messages with an unknown message type should
    be discarded", protocol.md:31).  Payloads: handshake (protocol, model) | audio bytes | text str | control int | metadata
    str | error str | ping None."""
    if not message:
        return None
    mt, payload = message[0], message[1:]
    if mt == MT_HANDSHAKE:
        if len(payload) >= 8:
            return mt, struct.unpack("<II", payload[:8])
        return mt, (PROTOCOL_VERSION, 0)             # the reference's Python server sends the bare type byte (server.py:166)
    if mt == MT_AUDIO:
        return mt, bytes(payload)
    if mt in (MT_TEXT, MT_METADATA, MT_ERROR):
        return mt, payload.decode("utf8", errors="replace")
    if mt == MT_CONTROL:
        return (mt, payload[0]) if payload else None
    if mt == MT_PING:
        return mt, None
    return None


# ---- audio codec seam ------------------------------------------------------------------------------------------------------------
class PcmCodec:
    """Raw PCM passthrough: the payload of an audio message is little-endian float32 samples (24 kHz mono).  Stands where the
    reference has `sphn.OpusStreamReader / OpusStreamWriter` (server.py:161-162)."""
    name = "pcm_f32le"

    class _Reader:
        def __init__(self):
            self._tail = b""

        def append_bytes(self, payload: bytes) -> np.ndarray:
            data = self._tail + payload
            n = len(data) // 4
            self._tail = data[4 * n:]
            return np.frombuffer(data[:4 * n], dtype="<f4").astype(np.float32)

    class _Writer:
        def append_pcm(self, pcm: np.ndarray) -> bytes:
            return np.ascontiguousarray(pcm, dtype="<f4").tobytes()

    def __init__(self, sample_rate: int = 24000):
        self.sample_rate = sample_rate

    def reader(self):
        return PcmCodec._Reader()

    def writer(self):
        return PcmCodec._Writer()


class OpusCodec:
    """Opus in Ogg pages through `sphn`, exactly the reference's objects.  Not available in images without sphn."""
    name = "opus"

    def __init__(self, sample_rate: int = 24000):
        try:
            import sphn  # noqa: F401
        except ImportError as e:                      # pragma: no cover - depends on the image
            raise RuntimeError("OpusCodec needs the `sphn` package (libopus); use PcmCodec in images without it") from e
        self._sphn = sphn
        self.sample_rate = sample_rate

    def reader(self):                                 # pragma: no cover
        return self._sphn.OpusStreamReader(self.sample_rate)

    def writer(self):                                 # pragma: no cover
        return self._sphn.OpusStreamWriter(self.sample_rate)


# ---- the server ------------------------------------------------------------------------------------------------------------------
@dataclass
class _Session:
    channel: int
    queue: "asyncio.Queue"
    loop: asyncio.AbstractEventLoop
    frames_out: int = 0
    closed: bool = False
    pending: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))


class BatchedServer:
    """`SessionBatcher` behind the websocket protocol.  `text_piece(token_id) -> str | None` turns a text token into what the
    reference sends (sentencepiece `id_to_piece` with "▁" -> " ", tokens 0 and 3 dropped, server.py:92-99); the default sends the
    token id.  `frame_period`: how long the model loop sleeps when no channel has a frame (the loop is frame-driven: it steps
    as soon as any channel holds a full 80 ms frame, like batched_asr.rs's model loop)."""

    def __init__(self, batcher, codec=None, text_piece: Optional[Callable[[int], Optional[str]]] = None, model_version: int = 0,
                 idle_sleep: float = 0.002):
        self.batcher = batcher
        self.codec = codec or PcmCodec()
        self.text_piece = text_piece or (lambda tok: None if tok in (0, 3) else str(int(tok)))
        self.model_version = model_version
        self.idle_sleep = idle_sleep
        self._sessions: Dict[int, _Session] = {}
        self._lock = threading.Lock()
        self._stop = threading.Event()
        self._thread: Optional[threading.Thread] = None
        self.errors = []

    @staticmethod
    def sentencepiece_text(tokenizer) -> Callable[[int], Optional[str]]:
        """The reference's text path (server.py:92-99) for a sentencepiece tokenizer."""
        def piece(tok: int) -> Optional[str]:
            if tok in (0, 3):
                return None
            return tokenizer.id_to_piece(int(tok)).replace("▁", " ")
        return piece

    # ---- model loop (one thread; the only caller of batcher.step) ------------------------------------------------------------
    def start(self) -> None:
        assert self._thread is None
        self._thread = threading.Thread(target=self._model_loop, name="moshi-model-loop", daemon=True)
        self._thread.start()

    def stop(self) -> None:
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=30)
            self._thread = None

    def _model_loop(self) -> None:
        try:
            while not self._stop.is_set():
                n = self.batcher.step()
                if n == 0:
                    time.sleep(self.idle_sleep)
                    continue
                with self._lock:
                    sessions = list(self._sessions.values())
                for sess in sessions:
                    if sess.closed:
                        continue
                    channel = sess.channel
                    while True:
                        # a handler may close (disconnect) or re-open (RESTART) the channel between the check above and the
                        # pop: a vanished channel is that session's business, not a failure of the model loop
                        try:
                            fr = self.batcher.pop(channel)
                        except UnknownChannel:
                            break
                        if fr is None or sess.closed or sess.channel != channel:
                            break
                        sess.loop.call_soon_threadsafe(sess.queue.put_nowait, fr)
        except BaseException as e:                    # noqa: BLE001 - surfaced to the handlers and the tests
            self.errors.append(e)
            with self._lock:
                for sess in self._sessions.values():
                    sess.loop.call_soon_threadsafe(sess.queue.put_nowait, e)

    # ---- one connection -------------------------------------------------------------------------------------------------------
    async def handle_chat(self, request):
        from aiohttp import web
        ws = web.WebSocketResponse()
        await ws.prepare(request)
        await self.serve_websocket(ws)
        return ws

    async def serve_websocket(self, ws) -> None:
        """Handshake, then audio in -> frames to the session's slot, frames out -> audio (+ text pieces).  A connection that
        finds every slot taken gets an Error message (MT=5) and is closed - the Rust server's behaviour; the reference's Python
        server would make it wait on the lock."""
        import aiohttp
        if self.errors:                               # the model loop is gone: a new connection would get a slot and no frames
            await ws.send_bytes(encode_error(f"model loop failed: {self.errors[0]!r}"))
            await ws.close()
            return
        try:
            channel = self.batcher.open()
        except BufferError as e:
            await ws.send_bytes(encode_error(f"no free slot: {e}"))
            await ws.close()
            return
        sess = _Session(channel, asyncio.Queue(), asyncio.get_running_loop())
        with self._lock:
            self._sessions[channel] = sess
        reader, writer = self.codec.reader(), self.codec.writer()
        F = self.batcher.frame_size

        async def send_loop():
            while True:
                item = await sess.queue.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    await ws.send_bytes(encode_error(f"model loop failed: {item!r}"))
                    await ws.close()
                    return
                pcm, tokens = item
                audio = writer.append_pcm(pcm)
                if len(audio) > 0:
                    await ws.send_bytes(encode_audio(audio))
                piece = self.text_piece(int(tokens[0]))
                if piece:
                    await ws.send_bytes(encode_text(piece))
                sess.frames_out += 1
        sender = asyncio.ensure_future(send_loop())
        try:
            await ws.send_bytes(encode_handshake(self.model_version))
            async for message in ws:
                if message.type != aiohttp.WSMsgType.BINARY:
                    if message.type in (aiohttp.WSMsgType.ERROR, aiohttp.WSMsgType.CLOSED):
                        break
                    continue                          # server.py:110-112: log and ignore
                msg = decode_message(message.data)
                if msg is None:
                    continue                          # empty / unknown kinds are discarded (protocol.md:31, server.py:148)
                kind, payload = msg
                if kind == MT_AUDIO:
                    pcm = reader.append_bytes(payload)
                    if pcm.shape[-1] == 0:
                        continue
                    sess.pending = np.concatenate((sess.pending, pcm))
                    n = (sess.pending.shape[0] // F) * F
                    if n:
                        self.batcher.push(channel, sess.pending[:n])
                        sess.pending = sess.pending[n:]
                elif kind == MT_CONTROL and payload == CONTROL_RESTART:
                    # a fresh dialogue on the same connection: the slot is released and re-opened (its rows are reset by the loop)
                    with self._lock:
                        self._sessions.pop(channel, None)
                    self.batcher.close(channel)
                    try:
                        channel = self.batcher.open()
                    except BufferError as e:          # another connection took the slot in between
                        await ws.send_bytes(encode_error(f"no free slot: {e}"))
                        await ws.close()
                        break
                    sess.channel = channel
                    sess.pending = np.zeros(0, np.float32)
                    while not sess.queue.empty():     # frames of the old dialogue that were not sent yet
                        sess.queue.get_nowait()
                    with self._lock:
                        self._sessions[channel] = sess
                elif kind == MT_PING:
                    continue
        finally:
            sess.closed = True
            with self._lock:
                self._sessions.pop(sess.channel, None)
            try:
                self.batcher.close(sess.channel)
            except Exception:                         # noqa: BLE001 - already closed by teardown
                pass
            sess.queue.put_nowait(None)
            await sender

    def make_app(self):
        from aiohttp import web
        app = web.Application()
        app.router.add_get("/api/chat", self.handle_chat)      # the reference's route (server.py:231)
        return app


def main(argv=None):                                  # pragma: no cover - needs a GPU and checkpoints
    """`python -m moshi_amd.server --checkpoint DIR`: the reference's `python -m moshi.server` for local checkpoints."""
    import argparse

    from aiohttp import web

    from .batcher import SessionBatcher
    from .loaders import CheckpointInfo
    ap = argparse.ArgumentParser()
    ap.add_argument("--host", default="localhost")
    ap.add_argument("--port", default=8998, type=int)
    ap.add_argument("--checkpoint", required=True, help="directory with config.json (CheckpointInfo.from_local)")
    ap.add_argument("--slots", type=int, default=32)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--cfg-coef", type=float, default=1.0)
    ap.add_argument("--codec", default="pcm", choices=["pcm", "opus"])
    args = ap.parse_args(argv)
    info = CheckpointInfo.from_local(args.checkpoint)
    mimi = info.get_mimi(device=args.device, max_batch=args.slots)
    lm = info.get_moshi(device=args.device, max_batch=args.slots * (2 if args.cfg_coef != 1.0 else 1))
    text_piece = None
    if info.tokenizer is not None and info.tokenizer.exists():
        import sentencepiece
        text_piece = BatchedServer.sentencepiece_text(sentencepiece.SentencePieceProcessor(str(info.tokenizer)))
    batcher = SessionBatcher(mimi, lm, args.slots, cfg_coef=args.cfg_coef, **info.lm_gen_config)
    server = BatchedServer(batcher, OpusCodec() if args.codec == "opus" else PcmCodec(), text_piece)
    server.start()
    try:
        web.run_app(server.make_app(), host=args.host, port=args.port)
    finally:
        server.stop()
        batcher.close_all()


if __name__ == "__main__":                            # pragma: no cover
    main()
