"""ctypes binding of libmoshi_mi.so (include/moshi_mi.h).

There is exactly one product library: `moshi_amd/libmoshi_mi.so`, built by hipcc for gfx950
(`python -m moshi_amd.build`).  If it is missing, importing the engine fails loudly - there is no
CPU or PyTorch fallback behind this API.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Dict, Optional, Sequence

import torch

_PKG = Path(__file__).resolve().parent
DEFAULT_LIB = _PKG / "libmoshi_mi.so"

MMI_OK, MMI_ERR_INVALID, MMI_ERR_SHAPE, MMI_ERR_STATE, MMI_ERR_HIP, MMI_ERR_MISSING_WEIGHT, MMI_ERR_UNSUPPORTED, MMI_ERR_BUSY, \
    MMI_ERR_NO_CHANNEL = 0, -1, -2, -3, -4, -5, -6, -7, -8
MMI_F32, MMI_BF16, MMI_I64, MMI_F16, MMI_I8, MMI_F8E4M3 = 0, 1, 2, 3, 4, 5

_DTYPES = {torch.float32: MMI_F32, torch.bfloat16: MMI_BF16, torch.int64: MMI_I64, torch.float16: MMI_F16,
           torch.int8: MMI_I8, torch.float8_e4m3fn: MMI_F8E4M3}


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("dtype", C.c_int32), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 4)]


class MimiCfg(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("frame_size", C.c_int32), ("channels", C.c_int32),
                ("dimension", C.c_int32), ("n_filters", C.c_int32), ("n_ratios", C.c_int32),
                ("ratios", C.c_int32 * 8), ("kernel_size", C.c_int32), ("last_kernel_size", C.c_int32),
                ("residual_kernel_size", C.c_int32), ("compress", C.c_int32), ("resample_stride", C.c_int32),
                ("tr_d_model", C.c_int32), ("tr_num_heads", C.c_int32), ("tr_num_layers", C.c_int32),
                ("tr_dim_feedforward", C.c_int32), ("tr_context", C.c_int32), ("tr_max_period", C.c_float),
                ("q_dimension", C.c_int32), ("q_bins", C.c_int32), ("q_n_q", C.c_int32),
                ("q_n_q_semantic", C.c_int32)]


class LMCfg(C.Structure):
    _fields_ = [("dim", C.c_int32), ("num_heads", C.c_int32), ("num_layers", C.c_int32), ("ffn_hidden", C.c_int32),
                ("context", C.c_int32), ("max_period", C.c_float), ("n_q", C.c_int32), ("dep_q", C.c_int32),
                ("card", C.c_int32), ("text_card", C.c_int32), ("text_card_out", C.c_int32),
                ("depformer_dim", C.c_int32), ("depformer_num_heads", C.c_int32),
                ("depformer_num_layers", C.c_int32), ("depformer_ffn_hidden", C.c_int32),
                ("delays", C.c_int32 * 64), ("existing_text_padding_id", C.c_int32),
                ("extra_heads_num_heads", C.c_int32), ("extra_heads_dim", C.c_int32), ("kv_cache_dtype", C.c_int32),
                ("cross_attention", C.c_int32)]


class Sampling(C.Structure):
    _fields_ = [("use_sampling", C.c_int32), ("temp", C.c_float), ("temp_text", C.c_float), ("top_k", C.c_int32),
                ("top_k_text", C.c_int32), ("seed", C.c_uint64)]


class Guidance(C.Structure):
    _fields_ = [("cfg_coef", C.c_float), ("cfg_is_no_text", C.c_int32), ("cfg_is_masked_until", C.c_void_p),
                ("condition_sum", C.c_void_p), ("condition_cross", C.c_void_p), ("cross_len", C.c_int32)]


HOOK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


class LMHooks(C.Structure):
    _fields_ = [("on_text_logits", HOOK_FN), ("on_text_token", HOOK_FN), ("on_audio_tokens", HOOK_FN), ("user", C.c_void_p)]


class BatcherCfg(C.Structure):
    _fields_ = [("slots", C.c_int32), ("reset_codec_after_first_frame", C.c_int32), ("max_buffered_frames", C.c_int32),
                ("sampling", Sampling), ("guidance", Guidance)]


class BatcherStats(C.Structure):
    _fields_ = [("steps", C.c_int64), ("frames", C.c_int64), ("dropped_frames", C.c_int64), ("used_slots", C.c_int32),
                ("total_slots", C.c_int32), ("last_step_ms", C.c_float)]


# name -> (restype, argtypes); every symbol include/moshi_mi.h declares
_P = C.c_void_p
SIGNATURES = {
    "mmi_version": (C.c_int, []),
    "mmi_last_error": (C.c_char_p, []),
    "mmi_mimi_create": (C.c_int, [C.POINTER(MimiCfg), C.POINTER(TensorDesc), C.c_int32, C.c_int32, C.POINTER(_P)]),
    "mmi_mimi_destroy": (None, [_P]),
    "mmi_mimi_set_num_codebooks": (C.c_int, [_P, C.c_int32]),
    "mmi_mimi_num_codebooks": (C.c_int, [_P]),
    "mmi_mimi_streaming_start": (C.c_int, [_P, C.c_int32, _P]),
    "mmi_mimi_streaming_stop": (C.c_int, [_P]),
    "mmi_mimi_set_exec_mask": (C.c_int, [_P, _P, _P]),
    "mmi_mimi_reset": (C.c_int, [_P, _P, _P]),
    "mmi_mimi_encode_step": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P]),
    "mmi_mimi_encode_latent_step": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P]),
    "mmi_mimi_quantize": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P]),
    "mmi_mimi_decode_latent": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "mmi_mimi_decode_step": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "mmi_mimi_decode_step_strided": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "mmi_mimi_streaming_batch": (C.c_int, [_P]),
    "mmi_lm_streaming_batch": (C.c_int, [_P]),
    "mmi_duplex_create": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "mmi_duplex_destroy": (None, [_P]),
    "mmi_duplex_submit": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.POINTER(C.c_int32), _P]),
    "mmi_duplex_batch": (C.c_int32, [_P]),
    "mmi_lm_has_hooks": (C.c_int32, [_P]),
    "mmi_lm_set_hidden_taps": (C.c_int, [_P, C.c_int32]),
    "mmi_lm_get_hidden_taps": (C.c_int, [_P, _P, C.c_int64, _P]),
    "mmi_duplex_join": (C.c_int, [_P, _P]),
    "mmi_duplex_flush": (C.c_int, [_P]),
    "mmi_duplex_set_timeline": (C.c_int, [_P, C.c_int32]),
    "mmi_duplex_get_timeline": (C.c_int, [_P, _P]),
    "mmi_duplex_get_stamps": (C.c_int, [_P, _P, _P]),
    "mmi_lm_create": (C.c_int, [C.POINTER(LMCfg), C.POINTER(TensorDesc), C.c_int32, C.c_int32, C.POINTER(_P)]),
    "mmi_lm_destroy": (None, [_P]),
    "mmi_lm_streaming_start": (C.c_int, [_P, C.c_int32, C.POINTER(Sampling), _P]),
    "mmi_lm_streaming_start_guided": (C.c_int, [_P, C.c_int32, C.POINTER(Sampling), C.POINTER(Guidance), _P]),
    "mmi_lm_model_rows": (C.c_int, [_P]),
    "mmi_lm_device": (C.c_int, [_P]),
    "mmi_mimi_device": (C.c_int, [_P]),
    "mmi_lm_stat": (C.c_int64, [_P, C.c_int32]),
    "mmi_lm_extra_heads": (C.c_int, [_P, _P, _P]),
    "mmi_lm_state_bytes": (C.c_int64, [_P]),
    "mmi_lm_state_save": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int64), _P]),
    "mmi_lm_state_load": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P]),
    "mmi_mimi_state_bytes": (C.c_int64, [_P]),
    "mmi_mimi_state_save": (C.c_int, [_P, _P, C.c_int64, _P]),
    "mmi_mimi_state_load": (C.c_int, [_P, _P, C.c_int64, _P]),
    "mmi_lm_streaming_stop": (C.c_int, [_P]),
    "mmi_lm_set_exec_mask": (C.c_int, [_P, _P, _P]),
    "mmi_lm_reset": (C.c_int, [_P, _P, _P]),
    "mmi_lm_step": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, _P, C.c_int32, C.POINTER(C.c_int32), _P]),
    "mmi_lm_force_next_tokens": (C.c_int, [_P, _P, _P]),
    "mmi_lm_set_phase_callback": (C.c_int, [_P, _P, _P]),
    "mmi_lm_set_hooks": (C.c_int, [_P, C.POINTER(LMHooks)]),
    "mmi_lm_hook_io": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_int64, _P]),
    "mmi_mimi_get_cfg": (C.c_int, [_P, C.POINTER(MimiCfg)]),
    "mmi_lm_get_cfg": (C.c_int, [_P, C.POINTER(LMCfg)]),
    "mmi_batcher_create": (C.c_int, [_P, _P, C.POINTER(BatcherCfg), C.POINTER(_P)]),
    "mmi_batcher_destroy": (None, [_P]),
    "mmi_batcher_open": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "mmi_batcher_close": (C.c_int, [_P, C.c_int64]),
    "mmi_batcher_push_pcm": (C.c_int, [_P, C.c_int64, _P, C.c_int32]),
    "mmi_batcher_step": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "mmi_batcher_pop": (C.c_int, [_P, C.c_int64, _P, _P, C.POINTER(C.c_int32)]),
    "mmi_batcher_get_stats": (C.c_int, [_P, C.POINTER(BatcherStats)]),
    "mmi_lm_debug_linear": (C.c_int, [_P, C.c_char_p, C.c_char_p, C.c_int32, _P, C.c_int32, _P, _P, _P, _P, _P]),
    "mmi_lm_launch_list": (C.c_int64, [_P, _P, C.c_int64]),
    "mmi_mimi_launch_list": (C.c_int64, [_P, C.c_int32, _P, C.c_int64]),
    "mmi_lm_seek": (C.c_int, [_P, _P, _P]),
    "mmi_lm_profile_begin": (C.c_int, [_P]),
    "mmi_lm_profile_sites": (C.c_int64, [_P, _P, C.c_int64]),
    "mmi_lm_profile_end": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_char_p)]),
}


class Lib:
    """A loaded engine library with typed entry points and the reference's error conventions."""

    def __init__(self, path: Path):
        self.path = Path(path)
        self.cdll = C.CDLL(str(self.path))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.cdll, name)  # AttributeError if the library does not export the ABI
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)

    def last_error(self) -> str:
        msg = self.mmi_last_error()
        return msg.decode() if msg else ""

    def check(self, rc: int) -> None:
        """Map a status to the exception type the reference raises in the same situation
        (RuntimeError for streaming misuse, AssertionError for shape/batch mismatch)."""
        if rc == MMI_OK:
            return
        msg = f"{self.last_error()} (mmi status {rc})"
        if rc == MMI_ERR_SHAPE:
            raise AssertionError(msg)
        if rc == MMI_ERR_INVALID:
            raise ValueError(msg)
        if rc == MMI_ERR_MISSING_WEIGHT:
            raise KeyError(msg)
        if rc == MMI_ERR_BUSY:
            raise BufferError(msg)
        if rc == MMI_ERR_NO_CHANNEL:                # its own status: no matching on the message text
            from .errors import UnknownChannel
            raise UnknownChannel(msg)
        if rc == MMI_ERR_UNSUPPORTED:
            raise NotImplementedError(msg)          # a RuntimeError, like every other engine failure
        raise RuntimeError(msg)


_default: Optional[Lib] = None


def load(path: Optional[Path] = None) -> Lib:
    """Load the engine.  With no argument: the in-tree gfx950 build, or a loud failure."""
    global _default
    if path is not None:
        return Lib(Path(path))
    if _default is None:
        override = os.environ.get("MMI_LIB_PATH")          # same-box A/B of two builds of the engine (scripts/gpu_*.sh)
        if override:
            _default = Lib(Path(override))
            return _default
        if not DEFAULT_LIB.exists():
            raise RuntimeError(
                f"{DEFAULT_LIB} is missing: build the HIP extension first (python -m moshi_amd.build). "
                "moshi_amd has no CPU/PyTorch fallback for this path.")
        _default = Lib(DEFAULT_LIB)
    return _default


def tensor_descs(state: Dict[str, torch.Tensor]):
    """Keep-alive list + C array of descriptors for a state dict (tensors must be contiguous)."""
    keep = []
    arr = (TensorDesc * len(state))()
    for i, (name, t) in enumerate(state.items()):
        if t.dtype not in _DTYPES:
            raise TypeError(f"unsupported dtype {t.dtype} for {name}")
        tc = t.contiguous()
        nm = name.encode()
        keep.append((tc, nm))
        arr[i].name = nm
        arr[i].data = tc.data_ptr()
        arr[i].dtype = _DTYPES[t.dtype]
        arr[i].ndim = tc.dim()
        if tc.dim() > 4:
            raise ValueError(f"rank > 4 tensor {name}")
        for d in range(tc.dim()):
            arr[i].shape[d] = tc.shape[d]
    return arr, keep


def stream_ptr(device: torch.device) -> Optional[int]:
    """hipStream_t of torch's current stream on `device` (None = the null stream, e.g. for host tensors)."""
    if device.type == "cuda":
        return torch.cuda.current_stream(device).cuda_stream
    return None


def read_text(call) -> str:
    """A `(buf, cap) -> bytes needed` text getter of the ABI, sized and read."""
    need = int(call(None, 0))
    if need <= 1:
        return ""
    buf = C.create_string_buffer(need)
    call(C.cast(buf, C.c_void_p), need)
    return buf.value.decode()


def launch_list(call, with_bytes: bool = False):
    """Decode a `mmi_*_launch_list` buffer: [(site, kernel)] in launch order ([(site, kernel, weight bytes)] with_bytes);
    [] before the first step."""
    out = []
    for line in read_text(call).splitlines():
        f = line.split("\t")
        site, kern, nb = f[0], f[1] if len(f) > 1 else "", int(f[2]) if len(f) > 2 else 0
        out.append((site, kern, nb) if with_bytes else (site, kern))
    return out


def device_scope(device):
    """Context in which `device` is the current HIP device: handles bind to the device current at `mmi_*_create` (every later
    entry point switches to it by itself, include/moshi_mi.h).  A no-op for the simulator's CPU device."""
    import contextlib
    import torch
    device = torch.device(device)
    if device.type != "cuda":
        return contextlib.nullcontext()
    return torch.cuda.device(device)
