"""`SessionBatcher` - many live full-duplex sessions on one GPU, host mirror of `mmi_batcher_*`.

The reference's Python server holds one session under a lock (moshi/moshi/server.py:45,57,154-169); its Rust server is the
model for batching: fixed slots, one channel per slot with a PCM FIFO, a model loop that each iteration takes one 80 ms
frame from every channel that has one, resets the rows of channels opened since, runs the batched step under that stream
mask and routes each row's result back (rust/moshi-server/src/batched_asr.rs:188-437, py_module.rs:443-470).  The loop
itself is native (moshi_amd/csrc/batcher.hip); this class only marshals host arrays.

    batcher = SessionBatcher(mimi, lm, slots=32)
    ch = batcher.open()                       # BufferError when every slot is taken
    batcher.push(ch, pcm_float32_24khz)       # any length; frames are cut by the batcher
    batcher.step()                            # one iteration of the model loop (call it from the loop thread)
    frame = batcher.pop(ch)                   # None or (pcm[1920] float32, tokens[1 + dep_q] int64)
    batcher.close(ch)
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _capi
from .errors import UnknownChannel  # noqa: F401  (re-exported: callers catch it from here)
from .lm import LMModel
from .mimi import MimiModel


class SessionBatcher:
    def __init__(self, mimi: MimiModel, lm_model: LMModel, slots: int, use_sampling: bool = True, temp: float = 0.8,
                 temp_text: float = 0.7, top_k: int = 250, top_k_text: int = 25, seed: int = 0,
                 reset_codec_after_first_frame: bool = True, max_buffered_frames: int = 250, cfg_coef: float = 1.0,
                 cfg_is_no_text: bool = False, cfg_is_masked_until=None, condition_tensors=None):
        assert mimi._lib is lm_model._lib, "both models must live in the same engine library"
        assert not mimi.is_streaming, "the batcher puts the models into streaming mode itself"
        self.mimi, self.lm_model = mimi, lm_model
        self._lib = mimi._lib
        self.frame_size = mimi.frame_size
        self.n_tokens = 1 + lm_model.dep_q
        cfg = _capi.BatcherCfg()
        cfg.slots = int(slots)
        cfg.reset_codec_after_first_frame = 1 if reset_codec_after_first_frame else 0
        cfg.max_buffered_frames = int(max_buffered_frames)
        cfg.sampling.use_sampling = 1 if use_sampling else 0
        cfg.sampling.temp, cfg.sampling.temp_text = temp, temp_text
        cfg.sampling.top_k, cfg.sampling.top_k_text = top_k, top_k_text
        cfg.sampling.seed = seed
        # guidance / conditioning shared by every channel (server.py:53-54 builds ONE set of condition tensors per model type)
        keep = []
        rows = int(slots) * (2 if cfg_coef != 1.0 else 1)
        cfg.guidance.cfg_coef = float(cfg_coef)
        cfg.guidance.cfg_is_no_text = 1 if cfg_is_no_text else 0
        if cfg_is_masked_until is not None and cfg_coef != 1.0:
            mu = (C.c_int64 * int(slots))(*[int(v) for v in cfg_is_masked_until])
            keep.append(mu)
            cfg.guidance.cfg_is_masked_until = C.cast(mu, C.c_void_p)
        if condition_tensors is not None:
            assert lm_model.fuser is not None, "Model has no fuser"
            cs = lm_model.fuser.get_sum(condition_tensors)
            if cs is not None:
                import torch
                assert cs.shape[0] == rows, "one condition row per model row (2 per slot when guided)"
                cs = cs.to(device=lm_model.device, dtype=torch.bfloat16).contiguous().view(rows, lm_model.dim)
                keep.append(cs)
                cfg.guidance.condition_sum = cs.data_ptr()
            cx = lm_model.fuser.get_cross(condition_tensors)
            if cx is not None:
                import torch
                assert cx.shape[0] == rows, "one condition row per model row (2 per slot when guided)"
                cx = cx.to(device=lm_model.device, dtype=torch.bfloat16).contiguous()
                keep.append(cx)
                cfg.guidance.condition_cross = cx.data_ptr()
                cfg.guidance.cross_len = int(cx.shape[1])
        assert mimi.device == lm_model.device, "the codec and the LM of a batcher must live on the same GPU"
        self._handle = C.c_void_p()
        mimi._sync()
        self._lib.check(self._lib.mmi_batcher_create(mimi._handle, lm_model._handle, C.byref(cfg), C.byref(self._handle)))

    def close_all(self) -> None:
        """Tear the batcher down (both models leave streaming mode)."""
        if getattr(self, "_handle", None) and self._handle.value:
            self._lib.mmi_batcher_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close_all()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close_all()

    # ---- channels ------------------------------------------------------------------------------------
    def open(self) -> int:
        ch = C.c_int64(0)
        self._lib.check(self._lib.mmi_batcher_open(self._handle, C.byref(ch)))
        return int(ch.value)

    def close(self, channel: int) -> None:
        self._lib.check(self._lib.mmi_batcher_close(self._handle, int(channel)))

    def push(self, channel: int, pcm) -> None:
        a = np.ascontiguousarray(np.asarray(pcm, dtype=np.float32).reshape(-1))
        self._lib.check(self._lib.mmi_batcher_push_pcm(self._handle, int(channel), a.ctypes.data, a.size))

    def pop(self, channel: int) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        """The channel's next output frame, or None.  A channel that is not (or no longer) open raises `UnknownChannel` - a
        ValueError like every MMI_ERR_INVALID, but one a model loop can tell apart from a real argument error."""
        pcm = np.empty(self.frame_size, dtype=np.float32)
        tok = np.empty(self.n_tokens, dtype=np.int64)
        got = C.c_int32(0)
        self._lib.check(self._lib.mmi_batcher_pop(self._handle, int(channel), pcm.ctypes.data, tok.ctypes.data, C.byref(got)))      # MMI_ERR_NO_CHANNEL -> UnknownChannel
        return (pcm, tok) if got.value else None

    # ---- model loop ----------------------------------------------------------------------------------
    def step(self) -> int:
        """One iteration of the model loop; returns the number of rows that had a frame (0 = nothing ran)."""
        n = C.c_int32(0)
        self._lib.check(self._lib.mmi_batcher_step(self._handle, C.byref(n)))
        return int(n.value)

    def stats(self) -> dict:
        s = _capi.BatcherStats()
        self._lib.check(self._lib.mmi_batcher_get_stats(self._handle, C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}

    @property
    def total_slots(self) -> int:      # py_module.rs:643-645
        return self.stats()["total_slots"]

    @property
    def used_slots(self) -> int:       # py_module.rs:647-649
        return self.stats()["used_slots"]
