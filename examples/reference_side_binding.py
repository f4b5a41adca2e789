"""The binding a maintainer of kyutai-labs/moshi would add to call libmoshi_mi.so, as a module that
RUNS: ctypes + torch only, nothing of `moshi_amd` is imported.  It would live in the reference as `moshi/models/_mi355x.py`; the
two classes take what the reference already has in hand - a model's `state_dict()` and the configuration dictionaries of
`moshi/moshi/models/loaders.py` (`_seanet_kwargs`, `_quantizer_kwargs`, `_transformer_kwargs` inside a `mimi_config`; `_lm_kwargs`)
- and expose the methods the reference's callers use (`server.py:59-72,135-147`, `run_inference.py:89-176`,
`scripts/moshi_benchmark.py:76-133`): `streaming_forever`, `streaming`, `encode`, `decode`, `step`, `reset_streaming`,
`set_exec_mask`.

    mimi = MimiModelMI355X(ref_mimi.state_dict(), mimi_config, max_batch=32)          # loaders.get_mimi's pieces
    gen = LMGenMI355X(ref_lm.state_dict(), lm_kwargs, max_batch=32, temp=0.8, temp_text=0.7)
    mimi.streaming_forever(32); gen.streaming_forever(32)
    tokens = gen.step(mimi.encode(pcm));  audio = mimi.decode(tokens[:, 1:]) if tokens is not None else None

tests/test_integration_stub.py runs it against `moshi_amd`'s own classes (bit-identical outputs) on the simulator build of the
library; on an MI355X the same file binds `moshi_amd/libmoshi_mi.so`."""
from __future__ import annotations

import ctypes as C
from contextlib import contextmanager
from pathlib import Path

import torch


# ---- include/moshi_mi.h, the structs this binding fills --------------------------------------------------------------------------
class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("dtype", C.c_int32), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


class MimiCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("sample_rate", "frame_size", "channels", "dimension", "n_filters", "n_ratios")] + \
               [("ratios", C.c_int32 * 8)] + \
               [(n, C.c_int32) for n in ("kernel_size", "last_kernel_size", "residual_kernel_size", "compress", "resample_stride", "tr_d_model",
                                         "tr_num_heads", "tr_num_layers", "tr_dim_feedforward", "tr_context")] + \
               [("tr_max_period", C.c_float)] + [(n, C.c_int32) for n in ("q_dimension", "q_bins", "q_n_q", "q_n_q_semantic")]


class LMCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim", "num_heads", "num_layers", "ffn_hidden", "context")] + [("max_period", C.c_float)] + \
               [(n, C.c_int32) for n in ("n_q", "dep_q", "card", "text_card", "text_card_out", "depformer_dim", "depformer_num_heads",
                                         "depformer_num_layers", "depformer_ffn_hidden")] + [("delays", C.c_int32 * 64)] + \
               [(n, C.c_int32) for n in ("existing_text_padding_id", "extra_heads_num_heads", "extra_heads_dim", "kv_cache_dtype", "cross_attention")]


class Sampling(C.Structure):
    _fields_ = [("use_sampling", C.c_int32), ("temp", C.c_float), ("temp_text", C.c_float), ("top_k", C.c_int32),
                ("top_k_text", C.c_int32), ("seed", C.c_uint64)]


_DT = {torch.float32: 0, torch.bfloat16: 1, torch.int64: 2, torch.float16: 3, torch.int8: 4}
_ERR = {-2: AssertionError, -3: RuntimeError, -5: KeyError, -1: ValueError}       # the reference's exception types
_P = C.c_void_p


def load(path: str | Path = "libmoshi_mi.so") -> C.CDLL:
    lib = C.CDLL(str(path))
    lib.mmi_last_error.restype = C.c_char_p
    for name, args in {
        "mmi_mimi_create": [C.POINTER(MimiCfg), C.POINTER(TensorDesc), C.c_int32, C.c_int32, C.POINTER(_P)],
        "mmi_mimi_destroy": [_P], "mmi_mimi_set_num_codebooks": [_P, C.c_int32], "mmi_mimi_streaming_start": [_P, C.c_int32, _P],
        "mmi_mimi_streaming_stop": [_P], "mmi_mimi_set_exec_mask": [_P, _P, _P], "mmi_mimi_reset": [_P, _P, _P],
        "mmi_mimi_encode_step": [_P, _P, _P, C.c_int32, C.c_int32, _P], "mmi_mimi_decode_step": [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P],
        "mmi_lm_create": [C.POINTER(LMCfg), C.POINTER(TensorDesc), C.c_int32, C.c_int32, C.POINTER(_P)], "mmi_lm_destroy": [_P],
        "mmi_lm_streaming_start": [_P, C.c_int32, C.POINTER(Sampling), _P], "mmi_lm_streaming_stop": [_P],
        "mmi_lm_set_exec_mask": [_P, _P, _P], "mmi_lm_reset": [_P, _P, _P],
        "mmi_lm_step": [_P, _P, C.c_int32, _P, _P, _P, _P, C.c_int32, C.POINTER(C.c_int32), _P],
    }.items():
        getattr(lib, name).argtypes = args
        getattr(lib, name).restype = None if name.endswith("destroy") else C.c_int
    return lib


def _check(lib, rc):
    if rc:
        raise _ERR.get(rc, RuntimeError)(f"{lib.mmi_last_error().decode()} (mmi status {rc})")


def _descs(state_dict):
    arr = (TensorDesc * len(state_dict))()
    for i, (k, v) in enumerate(state_dict.items()):
        arr[i].name, arr[i].data, arr[i].dtype, arr[i].ndim = k.encode(), v.data_ptr(), _DT[v.dtype], v.dim()
        for j, s in enumerate(v.shape):
            arr[i].shape[j] = s
    return arr


def _stream(device):                # enqueue on torch's current HIP stream: nothing synchronises (sampling.py:32-46)
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream) if device.type == "cuda" else None


def _mask_ptr(mask, device):
    if mask is None:
        return None, None
    m = mask.to(device=device, dtype=torch.uint8).contiguous()
    return m, m.data_ptr()


class _Streaming:
    """The part of StreamingModule the callers use (streaming.py:54-212)."""
    _batch = 0

    @property
    def is_streaming(self):
        return self._batch > 0

    @contextmanager
    def streaming(self, batch_size):
        self.streaming_forever(batch_size)
        try:
            yield
        finally:
            self._stop()


class MimiModelMI355X(_Streaming):              # compression.MimiModel
    def __init__(self, state_dict, mimi_config: dict, max_batch=32, num_codebooks=8, device="cuda", lib=None):
        self.device, self._lib = torch.device(device), lib or load()
        sea, q, tr = mimi_config["seanet"], mimi_config["quantizer"], mimi_config["transformer"]
        hop = 1
        for r in sea["ratios"]:
            hop *= r
        c = MimiCfg()
        c.sample_rate, c.channels = mimi_config["sample_rate"], mimi_config["channels"]
        c.frame_size = int(mimi_config["sample_rate"] / mimi_config["frame_rate"])
        c.dimension, c.n_filters, c.n_ratios = sea["dimension"], sea["n_filters"], len(sea["ratios"])
        for i, r in enumerate(sea["ratios"]):
            c.ratios[i] = r
        c.kernel_size, c.last_kernel_size, c.residual_kernel_size = sea["kernel_size"], sea["last_kernel_size"], sea["residual_kernel_size"]
        c.compress = sea["compress"]
        c.resample_stride = int(mimi_config["sample_rate"] / hop / mimi_config["frame_rate"])       # encoder rate / frame rate
        c.tr_d_model, c.tr_num_heads, c.tr_num_layers = tr["d_model"], tr["num_heads"], tr["num_layers"]
        c.tr_dim_feedforward, c.tr_context, c.tr_max_period = tr["dim_feedforward"], tr["context"], tr["max_period"]
        c.q_dimension, c.q_bins, c.q_n_q, c.q_n_q_semantic = q["dimension"], q["bins"], q["n_q"], 1
        self.frame_size, self.sample_rate, self.frame_rate = c.frame_size, c.sample_rate, mimi_config["frame_rate"]
        sd = {k: v.detach().to(device=self.device, dtype=torch.float32 if v.dtype.is_floating_point else v.dtype).contiguous()
              for k, v in state_dict.items()}
        self._h = _P()
        _check(self._lib, self._lib.mmi_mimi_create(C.byref(c), _descs(sd), len(sd), max_batch, C.byref(self._h)))
        self.num_codebooks = min(num_codebooks, q["n_q"])
        _check(self._lib, self._lib.mmi_mimi_set_num_codebooks(self._h, self.num_codebooks))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.mmi_mimi_destroy(self._h)

    def streaming_forever(self, batch_size):
        _check(self._lib, self._lib.mmi_mimi_streaming_start(self._h, batch_size, _stream(self.device)))
        self._batch = batch_size

    def _stop(self):
        self._lib.mmi_mimi_streaming_stop(self._h)
        self._batch = 0

    def reset_streaming(self, reset_mask=None):
        keep, p = _mask_ptr(reset_mask, self.device)
        _check(self._lib, self._lib.mmi_mimi_reset(self._h, p, _stream(self.device)))

    def set_exec_mask(self, exec_mask):
        keep, p = _mask_ptr(exec_mask, self.device)
        _check(self._lib, self._lib.mmi_mimi_set_exec_mask(self._h, p, _stream(self.device)))

    def encode(self, x):            # f32 [B, 1, frame_size * n] -> i64 [B, K, n]
        B, _, T = x.shape
        if T % self.frame_size:
            raise RuntimeError(f"streaming encode needs a multiple of the frame size {self.frame_size}")     # compression.py:361-365
        x = x.to(self.device, torch.float32).contiguous()
        codes = torch.empty(B, self.num_codebooks, T // self.frame_size, dtype=torch.int64, device=self.device)
        _check(self._lib, self._lib.mmi_mimi_encode_step(self._h, x.data_ptr(), codes.data_ptr(), B, T // self.frame_size, _stream(self.device)))
        return codes

    def decode(self, codes):        # i64 [B, K, n] -> f32 [B, 1, frame_size * n]
        B, K, n = codes.shape
        codes = codes.to(self.device, torch.int64).contiguous()
        pcm = torch.empty(B, 1, self.frame_size * n, dtype=torch.float32, device=self.device)
        _check(self._lib, self._lib.mmi_mimi_decode_step(self._h, codes.data_ptr(), pcm.data_ptr(), B, K, n, _stream(self.device)))
        return pcm


def _gating_hidden(dim, dim_feedforward):       # gating.py:55-58
    return (21 * dim) // 8 if dim_feedforward == 4 * dim else (2 * dim_feedforward) // 3


class LMGenMI355X(_Streaming):                  # lm.LMModel + lm.LMGen: the handle owns both
    def __init__(self, state_dict, lm_kwargs: dict, max_batch=32, use_sampling=True, temp=0.8, temp_text=0.7, top_k=250, top_k_text=25,
                 seed=0, device="cuda", lib=None):
        self.device, self._lib = torch.device(device), lib or load()
        k = lm_kwargs
        c = LMCfg()
        c.dim, c.num_heads, c.num_layers, c.context, c.max_period = k["dim"], k["num_heads"], k["num_layers"], k["context"], k["max_period"]
        c.ffn_hidden = _gating_hidden(k["dim"], int(k["hidden_scale"] * k["dim"]))
        c.n_q, c.dep_q, c.card, c.text_card, c.text_card_out = k["n_q"], k["dep_q"], k["card"], k["text_card"], k["text_card"]
        c.depformer_dim, c.depformer_num_heads, c.depformer_num_layers = k["depformer_dim"], k["depformer_num_heads"], k["depformer_num_layers"]
        c.depformer_ffn_hidden = _gating_hidden(k["depformer_dim"], k["depformer_dim_feedforward"])
        for i, d in enumerate(k["delays"]):
            c.delays[i] = d
        c.existing_text_padding_id, c.kv_cache_dtype = k.get("existing_text_padding_id", 3), 1          # MMI_BF16 ring
        self.dep_q, self.card, self.text_card, self.delays = k["dep_q"], k["card"], k["text_card"], list(k["delays"])
        self._needed = k["n_q"] - k["dep_q"]
        self._sampling = Sampling(int(use_sampling), temp, temp_text, top_k, top_k_text, seed)
        sd = {n: v.detach().to(device=self.device, dtype=torch.bfloat16 if v.dtype.is_floating_point else v.dtype).contiguous()
              for n, v in state_dict.items()}
        self._h = _P()
        _check(self._lib, self._lib.mmi_lm_create(C.byref(c), _descs(sd), len(sd), max_batch, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.mmi_lm_destroy(self._h)

    def streaming_forever(self, batch_size):
        _check(self._lib, self._lib.mmi_lm_streaming_start(self._h, batch_size, C.byref(self._sampling), _stream(self.device)))
        self._batch = batch_size

    def _stop(self):
        self._lib.mmi_lm_streaming_stop(self._h)
        self._batch = 0

    def reset_streaming(self, reset_mask=None):
        keep, p = _mask_ptr(reset_mask, self.device)
        _check(self._lib, self._lib.mmi_lm_reset(self._h, p, _stream(self.device)))

    def set_exec_mask(self, exec_mask):
        keep, p = _mask_ptr(exec_mask, self.device)
        _check(self._lib, self._lib.mmi_lm_set_exec_mask(self._h, p, _stream(self.device)))

    def step(self, input_tokens):   # i64 [B, >= n_q - dep_q, 1] -> i64 [B, 1 + dep_q, 1], or None while offset <= max_delay (lm.py:774-776)
        if not self.is_streaming:
            raise RuntimeError("You should wrap those calls with a `with lm_gen.streaming(): ...`.")          # lm.py:673-676
        B, K, S = input_tokens.shape
        assert S == 1 and K >= self._needed                                                                  # lm.py:679-686
        codes = input_tokens.to(self.device, torch.int64).contiguous()
        out = torch.empty(B, self.dep_q + 1, 1, dtype=torch.int64, device=self.device)
        valid = C.c_int32(0)
        _check(self._lib, self._lib.mmi_lm_step(self._h, codes.data_ptr(), K, out.data_ptr(), None, None, None, B, C.byref(valid),
                                                _stream(self.device)))
        return out if valid.value else None
