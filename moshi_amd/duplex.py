"""Software-pipelined frame step: `mimi.encode -> lm_gen.step -> mimi.decode` of consecutive frames overlapped on the GPU.

The reference's serving loop (moshi/moshi/server.py:132-146) runs

    codes = mimi.encode(chunk); tokens = lm_gen.step(codes); pcm = mimi.decode(tokens[:, 1:])

back to back on one stream.  `DuplexStream.step(chunk)` is the same three calls through `mmi_duplex_submit`
(include/moshi_mi.h): each model keeps its own stream order, so the results are bit-identical, but encode(t+1) and
decode(t-1) execute while LMGen.step(t) runs.  Outputs of frame t are valid on torch's current stream after `join()`.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _capi
from .lm import LMGen
from .mimi import MimiModel


class DuplexStream:
    """Both models must already be streaming with the same batch (`streaming_forever(B)` / inside `streaming(B)`).

    depth: output buffers kept alive (frames whose results may still be unread when `step` returns)."""

    def __init__(self, mimi: MimiModel, lm_gen: LMGen, depth: int = 4):
        assert depth >= 3, "two frames may be in flight when `step` returns"
        assert mimi._lib is lm_gen._lib, "both models must come from the same engine library"
        self._lib = mimi._lib
        self.mimi, self.lm_gen = mimi, lm_gen
        self.device = mimi.device
        h = C.c_void_p()
        with _capi.device_scope(self.device):
            self._lib.check(self._lib.mmi_duplex_create(mimi._handle, lm_gen.lm_model._handle, C.byref(h)))
        self._handle = h
        B = mimi._batch
        ntok = 1 + lm_gen.lm_model.dep_q
        self._pcm = [torch.zeros(B, mimi.channels, mimi.frame_size, device=self.device, dtype=torch.float32) for _ in range(depth)]
        self._tok = [torch.full((B, ntok, 1), -2, device=self.device, dtype=torch.int64) for _ in range(depth)]
        self._n = 0

    def close(self) -> None:
        if getattr(self, "_handle", None):
            self._lib.mmi_duplex_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, chunk: torch.Tensor, want_tokens: bool = True) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
        """Submit one 80 ms frame [B, C, frame_size].  Returns (tokens [B, 1 + dep_q, 1] or None, pcm [B, C, frame_size] or None)
        - None exactly where `lm_gen.step` returns None (lm.py:774-776).  The tensors are slots of a ring `depth` frames deep:
        read them after `join()` and before `depth` further steps.

        `chunk` is copied into the pipeline's own input ring in stream order (like `mimi.encode`, the caller may refill a
        preallocated input buffer as soon as this returns).  The batch must be the streaming batch (AssertionError, as
        `lm_gen.step`, lm.py:679-682).  An LMGen with per-step hooks is refused (NotImplementedError): hooks work on the caller's
        stream, the pipeline steps the LM on its own.  `support_out_of_sync` is honoured as `lm_gen.step` does (lm.py:774-776)."""
        x = self.mimi._check_audio(chunk)
        assert x.shape[-1] == self.mimi.frame_size, "one frame per step"
        assert x.shape[0] == self.mimi._batch, f"batch {x.shape[0]} != streaming batch {self.mimi._batch}"
        if self.lm_gen._hook_error is not None:          # a hook of an earlier serial step failed: surface it like lm_gen.step does
            err, self.lm_gen._hook_error = self.lm_gen._hook_error, None
            raise err
        slot = self._n % len(self._pcm)
        self._n += 1
        pcm, tok = self._pcm[slot], self._tok[slot]
        valid = C.c_int32(0)
        self._lib.check(self._lib.mmi_duplex_submit(self._handle, x.data_ptr(), pcm.data_ptr(), tok.data_ptr() if want_tokens else None,
                                                    x.shape[0], C.byref(valid), _capi.stream_ptr(self.device)))
        if not valid.value:      # inside the LM's delay nothing is decoded; `support_out_of_sync` still hands the -2 rows out
            return ((tok if want_tokens else None), None) if self.lm_gen.support_out_of_sync else (None, None)
        return (tok if want_tokens else None), pcm

    def join(self) -> None:
        """torch's current stream waits (on the device) for every submitted frame."""
        self._lib.check(self._lib.mmi_duplex_join(self._handle, _capi.stream_ptr(self.device)))

    def flush(self) -> None:
        """The host waits until every submitted frame has completed (mmi_duplex_flush)."""
        self._lib.check(self._lib.mmi_duplex_flush(self._handle))

    def timeline(self, on: Optional[bool] = None):
        """Diagnostics (mmi_duplex_set_timeline / get_timeline): `timeline(True)` switches the per-phase timestamps on;
        `timeline()` synchronises and returns the last frame's {phase: (begin_ms, end_ms)} since its submit."""
        if on is not None:
            self._lib.check(self._lib.mmi_duplex_set_timeline(self._handle, 1 if on else 0))
            return None
        ms = (C.c_float * 7)()
        self._lib.check(self._lib.mmi_duplex_get_timeline(self._handle, C.cast(ms, C.c_void_p)))
        return {"encode": (ms[0], ms[1]), "lm": (ms[2], ms[4]), "decode": (ms[5], ms[6])}

    STAMPS = ("in", "enc0", "enc1", "wait0", "wait1", "lm0", "phase", "lm1", "dec0", "dec1")

    def stamps(self):
        """Diagnostics (mmi_duplex_get_stamps): device-clock stamps of the last four frames, {frame: {point: ms}} on one time base."""
        ms = (C.c_double * 40)()
        last = C.c_int64(0)
        self._lib.check(self._lib.mmi_duplex_get_stamps(self._handle, C.cast(ms, C.c_void_p), C.byref(last)))
        out = {}
        for f in range(max(0, last.value - 3), last.value + 1):
            out[f] = {n: ms[(f & 3) * 10 + i] for i, n in enumerate(self.STAMPS) if ms[(f & 3) * 10 + i] >= 0}
        return out
