"""`MimiModel` - host-side mirror of the reference's Mimi API on top of the HIP engine.

Same method names, argument meaning, shapes, dtypes and error behaviour as the reference's
`MimiModel` + `StreamingModule` (reference: moshi/moshi/models/compression.py:105-433,
moshi/moshi/modules/streaming.py:54-212), so a caller such as `server.py:59-72,135,144`
or `run_inference.py:89-90,141,176` can switch without changes.  All arithmetic happens in
libmoshi_mi.so (hand-written gfx950 kernels); this file only owns tensors and lifetimes.
"""
from __future__ import annotations

import ctypes as C
from contextlib import contextmanager
from typing import Dict, Optional

import torch

from . import _capi
from .config import MimiConfig


def _mimi_cfg_struct(cfg: MimiConfig) -> _capi.MimiCfg:
    s = _capi.MimiCfg()
    s.sample_rate = cfg.sample_rate
    s.frame_size = cfg.frame_size
    s.channels = cfg.channels
    s.dimension = cfg.dimension
    s.n_filters = cfg.n_filters
    s.n_ratios = len(cfg.ratios)
    for i, r in enumerate(cfg.ratios):
        s.ratios[i] = r
    s.kernel_size = cfg.kernel_size
    s.last_kernel_size = cfg.last_kernel_size
    s.residual_kernel_size = cfg.residual_kernel_size
    s.compress = cfg.compress
    s.resample_stride = cfg.resample_stride
    s.tr_d_model = cfg.tr_d_model
    s.tr_num_heads = cfg.tr_num_heads
    s.tr_num_layers = cfg.tr_num_layers
    s.tr_dim_feedforward = cfg.tr_dim_feedforward
    s.tr_context = cfg.tr_context
    s.tr_max_period = cfg.tr_max_period
    s.q_dimension = cfg.q_dimension
    s.q_bins = cfg.q_bins
    s.q_n_q = cfg.q_n_q
    s.q_n_q_semantic = cfg.q_n_q_semantic
    return s


class MimiModel:
    """Mimi codec running on the MI355X engine.

    Args:
        state_dict: tensors named as in the reference checkpoints, fp32.
        config: architecture hyper-parameters.
        device: where the tensors and the engine live (a ROCm `cuda` device for the product library).
        max_batch: largest streaming batch this instance will be asked for.
        num_codebooks: active codebooks (reference: `get_mimi(..., num_codebooks=8)`, loaders.py:362).
        lib: engine library; default = the in-tree gfx950 build (tests inject the kernel simulator here).
    """

    def __init__(self, state_dict: Dict[str, torch.Tensor], config: Optional[MimiConfig] = None,
                 device: torch.device | str = "cuda", max_batch: int = 64, num_codebooks: int = 8,
                 lib: Optional[_capi.Lib] = None):
        self.config = config or MimiConfig()
        self.device = torch.device(device)
        if lib is None:
            if self.device.type != "cuda":
                raise RuntimeError("moshi_amd.MimiModel runs on an MI355X (device='cuda'); there is no CPU path")
            lib = _capi.load()
        self._lib = lib
        self._handle = C.c_void_p()
        self._batch: Optional[int] = None
        from .weights import normalize_mimi_state_dict
        state_dict = normalize_mimi_state_dict(state_dict)               # legacy codebook / fused-projection key names
        sd = {k: v.detach().to(device=self.device, dtype=torch.float32) for k, v in state_dict.items()
              if v.dtype.is_floating_point}
        descs, keep = _capi.tensor_descs(sd)
        cfg = _mimi_cfg_struct(self.config)
        self._sync()
        with _capi.device_scope(self.device):             # the handle binds to the device current at create
            lib.check(lib.mmi_mimi_create(C.byref(cfg), descs, len(sd), max_batch, C.byref(self._handle)))
        del keep
        self.max_batch = max_batch
        self.set_num_codebooks(min(num_codebooks, self.config.q_n_q))

    # ---- plumbing ------------------------------------------------------------------------------
    def _sync(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def _stream(self):
        return _capi.stream_ptr(self.device)

    def __del__(self):
        try:
            if getattr(self, "_handle", None) and self._handle.value:
                self._sync()
                self._lib.mmi_mimi_destroy(self._handle)
                self._handle = C.c_void_p()
        except Exception:
            pass

    def _mask_arg(self, mask: Optional[torch.Tensor]):
        if mask is None:
            return None, None
        m = mask.to(device=self.device, dtype=torch.bool).contiguous().view(torch.uint8)
        assert m.numel() == self._batch, f"mask has {m.numel()} entries, streaming batch is {self._batch}"
        return m, m.data_ptr()

    def launch_list(self, which: str = "encode"):
        """[(site, kernel)] per kernel launch of one encoder / decoder step (recorded during the first step)."""
        w = {"encode": 0, "decode": 1}[which]
        return _capi.launch_list(lambda buf, cap: self._lib.mmi_mimi_launch_list(self._handle, w, buf, cap))

    # ---- properties (compression.py:232-265) -----------------------------------------------------
    @property
    def channels(self) -> int:
        return self.config.channels

    @property
    def frame_rate(self) -> float:
        return self.config.frame_rate

    @property
    def sample_rate(self) -> int:
        return self.config.sample_rate

    @property
    def frame_size(self) -> int:
        return self.config.frame_size

    @property
    def dimension(self) -> int:
        return self.config.dimension

    @property
    def total_codebooks(self) -> int:
        return self.config.q_n_q

    @property
    def num_codebooks(self) -> int:
        return int(self._lib.mmi_mimi_num_codebooks(self._handle))

    @property
    def cardinality(self) -> int:
        return self.config.q_bins

    def set_num_codebooks(self, n: int) -> None:
        self._lib.check(self._lib.mmi_mimi_set_num_codebooks(self._handle, int(n)))

    # ---- streaming lifecycle (streaming.py:110-211) ----------------------------------------------
    @property
    def is_streaming(self) -> bool:
        return self._batch is not None

    def streaming_forever(self, batch_size: int) -> None:
        self._lib.check(self._lib.mmi_mimi_streaming_start(self._handle, int(batch_size), self._stream()))
        self._batch = int(batch_size)

    def _stop_streaming(self) -> None:
        self._sync()
        self._lib.check(self._lib.mmi_mimi_streaming_stop(self._handle))
        self._batch = None

    @contextmanager
    def streaming(self, batch_size: int):
        self.streaming_forever(batch_size)
        try:
            yield self
        finally:
            self._stop_streaming()

    def reset_streaming(self, reset_mask: Optional[torch.Tensor] = None) -> None:
        assert self.is_streaming, "Trying to reset streaming, but the model wasn't streaming."
        keep, ptr = self._mask_arg(reset_mask)
        self._lib.check(self._lib.mmi_mimi_reset(self._handle, ptr, self._stream()))

    def set_streaming_detached(self, streaming_detached: bool) -> None:
        """streaming.py:78-86: keeps a parent module's `.streaming()` from reaching this one.  An engine handle has no parent and
        enters streaming mode only through its own `streaming()` / `streaming_forever()`: always detached; the flag is kept."""
        self._streaming_detached = bool(streaming_detached)

    def set_exec_mask(self, exec_mask: torch.Tensor) -> None:
        assert self.is_streaming
        keep, ptr = self._mask_arg(exec_mask)
        self._lib.check(self._lib.mmi_mimi_set_exec_mask(self._handle, ptr, self._stream()))

    def get_streaming_state(self) -> dict:
        """streaming.py:158-166: the complete streaming state (a copy: one opaque device tensor)."""
        assert self.is_streaming
        n = int(self._lib.mmi_mimi_state_bytes(self._handle))
        buf = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._lib.check(self._lib.mmi_mimi_state_save(self._handle, buf.data_ptr(), n, self._stream()))
        return {"mimi": buf, "batch_size": self._batch}

    def set_streaming_state(self, state: dict) -> None:
        """streaming.py:168-181."""
        assert self.is_streaming
        if "mimi" not in state:
            raise RuntimeError("Expected to find a streaming state for mimi.")
        buf = state["mimi"]
        self._lib.check(self._lib.mmi_mimi_state_load(self._handle, buf.data_ptr(), buf.numel(), self._stream()))

    # ---- encode / decode (compression.py:338-433) ------------------------------------------------
    def _check_audio(self, x: torch.Tensor) -> torch.Tensor:
        assert x.dim() == 3, f"expects audio of shape [B, C, T] but got {tuple(x.shape)}"
        assert x.shape[1] == self.channels
        return x.to(device=self.device, dtype=torch.float32).contiguous()

    def _frames_streaming(self, x: torch.Tensor) -> int:
        fs = self.frame_size
        if x.shape[-1] % fs != 0 or x.shape[-1] == 0:
            raise RuntimeError(
                f"Invalid input x of length {x.shape[-1]}. The length must be a positive multiple of the frame "
                f"size {fs}. You are responsible for buffering accordingly before feeding audio to Mimi.")
        return x.shape[-1] // fs

    def _encode_impl(self, x: torch.Tensor, want_latent: bool) -> torch.Tensor:
        x = self._check_audio(x)
        temporary = not self.is_streaming
        if temporary:
            # non-streaming call == streaming from a fresh state over the right-padded signal (compression.py:354-359)
            fs = self.frame_size
            pad = (-x.shape[-1]) % fs
            if pad:
                x = torch.nn.functional.pad(x, (0, pad))
            self.streaming_forever(x.shape[0])
        try:
            B = x.shape[0]
            n = self._frames_streaming(x)
            if want_latent:
                out = torch.empty(B, self.dimension, n, device=self.device, dtype=torch.float32)
                rc = self._lib.mmi_mimi_encode_latent_step(self._handle, x.data_ptr(), out.data_ptr(), B, n, self._stream())
            else:
                out = torch.empty(B, self.num_codebooks, n, device=self.device, dtype=torch.int64)
                rc = self._lib.mmi_mimi_encode_step(self._handle, x.data_ptr(), out.data_ptr(), B, n, self._stream())
            self._lib.check(rc)
        finally:
            if temporary:
                self._stop_streaming()
        return out

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """[B, C, T] float -> [B, K, T / frame_size] int64 codes."""
        return self._encode_impl(x, want_latent=False)

    def encode_to_latent(self, x: torch.Tensor, quantize: bool = True) -> torch.Tensor:
        if not quantize:
            return self._encode_impl(x, want_latent=True)
        return self.decode_latent(self.encode(x))

    def quantize(self, latent: torch.Tensor) -> torch.Tensor:
        """`quantizer.encode` on a given latent [B, dimension, T] -> codes [B, K, T] (vq.py:269-279)."""
        lat = latent.to(device=self.device, dtype=torch.float32).contiguous()
        B, D, n = lat.shape
        assert D == self.dimension
        out = torch.empty(B, self.num_codebooks, n, device=self.device, dtype=torch.int64)
        self._lib.check(self._lib.mmi_mimi_quantize(self._handle, lat.data_ptr(), out.data_ptr(), B, n, self._stream()))
        return out

    def decode_latent(self, codes: torch.Tensor) -> torch.Tensor:
        codes = codes.to(device=self.device, dtype=torch.int64).contiguous()
        B, K, n = codes.shape
        out = torch.empty(B, self.dimension, n, device=self.device, dtype=torch.float32)
        self._lib.check(self._lib.mmi_mimi_decode_latent(self._handle, codes.data_ptr(), out.data_ptr(), B, K, n,
                                                         self._stream()))
        return out

    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        """[B, K, T'] int64 codes -> [B, C, T' * frame_size] float."""
        assert codes.dim() == 3
        codes = codes.to(device=self.device, dtype=torch.int64)
        B, K, n = codes.shape
        # a column slice of a wider tensor - `tokens[:, 1:]` of LMGen.step's output (server.py:144-146) - is read in place
        strided = codes.stride(2) == 1 and codes.stride(1) == n and codes.stride(0) >= K * n
        if not strided:
            codes = codes.contiguous()
        temporary = not self.is_streaming
        if temporary:
            self.streaming_forever(B)
        try:
            out = torch.empty(B, self.channels, n * self.frame_size, device=self.device, dtype=torch.float32)
            self._lib.check(self._lib.mmi_mimi_decode_step_strided(self._handle, codes.data_ptr(), codes.stride(0), out.data_ptr(),
                                                                   B, K, n, self._stream()))
        finally:
            if temporary:
                self._stop_streaming()
        return out
