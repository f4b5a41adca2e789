"""`LMModel` / `LMGen` - host-side mirror of the reference's Moshi LM generation API on the HIP engine.

Same constructor arguments, method names, shapes, dtypes, `None`-during-delay behaviour and exception types as
the reference (moshi/moshi/models/lm.py:49-320 `LMModel` attributes, :556-850 `LMGen`), so that callers such as
`server.py:59-72,144` or `run_inference.py:89-90,160-171` can switch.  All arithmetic happens in libmoshi_mi.so.
"""
from __future__ import annotations

import ctypes as C
from contextlib import contextmanager
from typing import Dict, List, Optional, Tuple

import torch

from . import _capi
from .config import LMConfig


def _lm_cfg_struct(cfg: LMConfig) -> _capi.LMCfg:
    s = _capi.LMCfg()
    s.dim = cfg.dim
    s.num_heads = cfg.num_heads
    s.num_layers = cfg.num_layers
    s.ffn_hidden = cfg.ffn_hidden
    s.context = cfg.context
    s.max_period = cfg.max_period
    s.n_q = cfg.n_q
    s.dep_q = cfg.dep_q
    s.card = cfg.card
    s.text_card = cfg.text_card
    s.text_card_out = cfg.text_card
    s.depformer_dim = cfg.depformer_dim
    s.depformer_num_heads = cfg.depformer_num_heads
    s.depformer_num_layers = cfg.depformer_num_layers
    s.depformer_ffn_hidden = cfg.depformer_ffn_hidden
    assert len(cfg.delays) == cfg.n_q + 1, f"expected {cfg.n_q + 1} delays, got {len(cfg.delays)}."
    for i, d in enumerate(cfg.delays):
        s.delays[i] = d
    s.existing_text_padding_id = cfg.existing_text_padding_id
    s.extra_heads_num_heads = cfg.extra_heads_num_heads
    s.extra_heads_dim = cfg.extra_heads_dim
    if cfg.kv_cache_dtype not in ("bf16", "fp8"):
        raise ValueError("kv_cache_dtype must be 'bf16' or 'fp8'")
    s.kv_cache_dtype = _capi.MMI_F8E4M3 if cfg.kv_cache_dtype == "fp8" else _capi.MMI_BF16
    s.cross_attention = 1 if cfg.cross_attention else 0
    return s


class ConditionFuser:
    """The part of the reference's `ConditionFuser` that `LMGen` uses (conditioners/base.py:349-421): which named condition
    tensors are summed into the model input (`sum`) and which are concatenated along time into the source every temporal
    layer cross-attends to (`cross`).  `prepend` is not implemented by the reference's generation path either."""

    def __init__(self, fuse2cond: Dict[str, List[str]], cross_attention_pos_emb: bool = False,
                 cross_attention_pos_emb_scale: float = 1.0):
        for method, names in fuse2cond.items():
            if method not in ("sum", "cross", "prepend"):
                raise AssertionError(f"Got invalid fuse method {method}")
            if method == "prepend" and names:
                raise RuntimeError(f"only `sum` and `cross` conditionings are supported for now, got {method}.")   # base.py:380-381
        self.fuse2cond = {"sum": list(fuse2cond.get("sum", [])), "cross": list(fuse2cond.get("cross", [])), "prepend": []}
        self.cross_attention_pos_emb = cross_attention_pos_emb
        self.cross_attention_pos_emb_scale = cross_attention_pos_emb_scale

    def get_sum(self, conditions) -> Optional[torch.Tensor]:
        """conditioners/base.py:410-421: conditions[name] = (tensor [B, 1, dim], mask)."""
        total = None
        for name in self.fuse2cond["sum"]:
            cond = conditions[name][0]
            assert cond.shape[1] == 1, cond.shape
            total = cond if total is None else total + cond
        return total

    def get_cross(self, conditions) -> Optional[torch.Tensor]:
        """conditioners/base.py:392-409: the `cross` conditions concatenated along time, [B, T_c, dim] (+ a sinusoidal
        position embedding when the fuser was built with `cross_attention_pos_emb`)."""
        cross = None
        for name in self.fuse2cond["cross"]:
            cond = conditions[name][0]
            cross = cond if cross is None else torch.cat([cross, cond], dim=1)
        if self.cross_attention_pos_emb and cross is not None:
            positions = torch.arange(cross.shape[1], device=cross.device).view(1, -1, 1)
            half = cross.shape[-1] // 2                     # modules/transformer.py:139-164 create_sin_embedding
            adim = torch.arange(half, device=cross.device, dtype=torch.float32).view(1, 1, -1)
            phase = positions.to(torch.float32) / (torch.full([], 10000.0, device=cross.device) ** (adim / (half - 1)))
            pos_emb = torch.cat([torch.cos(phase), torch.sin(phase)], dim=-1).to(cross.dtype)
            cross = cross + self.cross_attention_pos_emb_scale * pos_emb
        return cross


class LMModel:
    """Weights + architecture of the Moshi LM on the engine (reference: lm.py:49-320).

    Args:
        state_dict: bf16 tensors named as in the reference checkpoints.
        config: architecture hyper-parameters (defaults = Moshi-7B, loaders.py:90-119).
        device: a ROCm `cuda` device for the product library.
        max_batch: largest number of concurrent sessions `LMGen.streaming` will be asked for.
        quantize: True converts the linears to row-wise int8 (`weight` + `weight_scb`), like the reference's `quantize=True`;
            "fp8" converts them to e4m3fn (`weight` + `weight_scale`) for the fp8 MFMA path (BASELINE configs[4]);
            a state dict that already carries int8 / fp8 weights is used as is.
    """

    def __init__(self, state_dict: Dict[str, torch.Tensor], config: Optional[LMConfig] = None,
                 device: torch.device | str = "cuda", max_batch: int = 32, lib: Optional[_capi.Lib] = None,
                 quantize: bool | str = False, fuser: Optional[ConditionFuser] = None, kv_cache: Optional[str] = None):
        self.config = config or LMConfig()
        if kv_cache is not None:         # "fp8": e4m3 KV ring (half the attention stream); default: the config's (bf16)
            from dataclasses import replace
            self.config = replace(self.config, kv_cache_dtype=kv_cache)
        self.fuser = fuser
        self.device = torch.device(device)
        if lib is None:
            if self.device.type != "cuda":
                raise RuntimeError("moshi_amd.LMModel runs on an MI355X (device='cuda'); there is no CPU path")
            lib = _capi.load()
        self._lib = lib
        self._handle = C.c_void_p()
        from .weights import normalize_lm_state_dict, quantize_lm_state_dict, quantize_lm_state_dict_fp8
        state_dict = normalize_lm_state_dict(state_dict, self.config)     # fused multi-step projections of released checkpoints
        if quantize == "fp8":
            state_dict = quantize_lm_state_dict_fp8(state_dict)
        elif quantize:                                                    # the reference's `quantize=True` (lm.py:242-243)
            state_dict = quantize_lm_state_dict(state_dict)
        self.quantized = any(v.dtype in (torch.int8, torch.float8_e4m3fn) for v in state_dict.values())
        if self.quantized and getattr(config, "cross_attention", False):
            # the engine runs one weight format per model (activation buffers are laid out for it) and keeps cross-attention
            # linears bf16: fail here, with the reason, instead of inside mmi_lm_create
            raise NotImplementedError("quantised linears (int8 / fp8) are not supported for models with cross-attention layers; "
                                      "load the bf16 checkpoint (quantize=False)")

        def place(k, v):     # quantised weights and their fp32 scales keep their dtype (utils/quantize.py:29-34); the rest is bf16
            if v.dtype in (torch.int8, torch.float8_e4m3fn):
                return v.detach().to(self.device)
            scale = k.endswith("_scb") or k.endswith(".weight_scale") or k.endswith(".input_scale")
            return v.detach().to(device=self.device, dtype=torch.float32 if scale else torch.bfloat16)
        sd = {k: place(k, v) for k, v in state_dict.items()}
        descs, keep = _capi.tensor_descs(sd)
        cfg = _lm_cfg_struct(self.config)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        with _capi.device_scope(self.device):             # the handle binds to the device current at create
            lib.check(lib.mmi_lm_create(C.byref(cfg), descs, len(sd), max_batch, C.byref(self._handle)))
        del keep, sd
        self.max_batch = max_batch
        self.training = False

    def __del__(self):
        try:
            if getattr(self, "_handle", None) and self._handle.value:
                if self.device.type == "cuda":
                    torch.cuda.synchronize(self.device)
                self._lib.mmi_lm_destroy(self._handle)
                self._handle = C.c_void_p()
        except Exception:
            pass

    def enable_hidden_taps(self, on: bool = True) -> None:
        """Parity tap (tests): every step of the NEXT `LMGen.streaming()` also keeps the residual stream after the first and
        the last temporal layer (`LMGen.hidden_taps()`)."""
        self._lib.check(self._lib.mmi_lm_set_hidden_taps(self._handle, 1 if on else 0))

    DEBUG_PATHS = {"plain": 0, "splitk": 1, "fused": 2, "norm": 3, "norm_fused": 4}

    def debug_linear(self, weight_name: str, x: torch.Tensor, path: str = "plain", alpha_name: Optional[str] = None,
                     want_codes: bool = False) -> dict:
        """Parity tap (tests; `mmi_lm_debug_linear`): ONE linear of the model - the module stored under `weight_name`, i.e.
        `F.linear` / `QLinear.forward` (utils/quantize.py:24-40) - on the bf16 rows `x` [rows, in_features], through the kernels
        the step uses for it.  Returns {"out": bf16 [rows, out_features] (a gated linear_in: silu(gate) * value), and on request
        "codes" int8 [rows, in_features] / "absmax" fp32 [rows] (the row-wise quantisation the int8 x int8 GEMM consumed),
        "norm": the normalised rows (path "norm")}."""
        lib = self._lib
        x = x.to(device=self.device, dtype=torch.bfloat16).contiguous()
        rows, K = x.shape
        # the C entry sizes its operand from the packed weight: rows of another width would be read and written out of bounds
        assert K == self._linear_in_features(weight_name), f"{weight_name} takes rows of {self._linear_in_features(weight_name)} features, got {K}"
        w = weight_name.encode()
        # out_features from the config-independent side: ask for a generous buffer, the engine writes [rows][N]
        n_out = self._linear_out_features(weight_name)
        out = torch.empty(rows, n_out, dtype=torch.bfloat16, device=self.device)
        codes = torch.empty(rows, K, dtype=torch.int8, device=self.device) if want_codes else None
        absmax = torch.empty(rows, dtype=torch.float32, device=self.device) if want_codes else None
        norm = torch.empty(rows, K, dtype=torch.bfloat16, device=self.device) if path == "norm" else None
        ptr = lambda t: None if t is None else t.data_ptr()
        lib.check(lib.mmi_lm_debug_linear(self._handle, w, alpha_name.encode() if alpha_name else None, self.DEBUG_PATHS[path],
                                          x.data_ptr(), rows, out.data_ptr(), ptr(codes), ptr(absmax), ptr(norm),
                                          _capi.stream_ptr(self.device)))
        res = {"out": out}
        if want_codes:
            res.update(codes=codes, absmax=absmax)
        if norm is not None:
            res["norm"] = norm
        return res

    def _linear_in_features(self, weight_name: str) -> int:
        c = self.config
        dep = weight_name.startswith("depformer.") or weight_name.startswith("linears.")
        if "linear_out" in weight_name:
            return c.depformer_ffn_hidden if dep else c.ffn_hidden
        return c.depformer_dim if dep else c.dim          # (depformer_in.* and text_linear read the temporal transformer's output)

    def _linear_out_features(self, weight_name: str) -> int:
        c = self.config
        dd = c.depformer_dim
        dep = weight_name.startswith("depformer.")
        d = dd if dep else c.dim
        if ".self_attn.in_projs." in weight_name:
            return 3 * d
        if ".self_attn.out_projs." in weight_name:
            return d
        if "linear_in" in weight_name:           # gated: the engine returns the hidden tensor
            return c.depformer_ffn_hidden if dep else c.ffn_hidden
        if "linear_out" in weight_name:
            return d
        if weight_name.startswith("text_linear"):
            return c.text_card
        if weight_name.startswith("depformer_in."):
            return dd
        if weight_name.startswith("linears."):
            return c.card
        raise KeyError(weight_name)

    # attributes callers read
    @property
    def dep_q(self) -> int:
        return self.config.dep_q

    @property
    def n_q(self) -> int:
        return self.config.n_q

    @property
    def card(self) -> int:
        return self.config.card

    @property
    def text_card(self) -> int:
        return self.config.text_card

    @property
    def delays(self) -> List[int]:
        return list(self.config.delays)

    @property
    def dim(self) -> int:
        return self.config.dim

    @property
    def dtype(self) -> torch.dtype:
        return torch.bfloat16

    @property
    def num_codebooks(self) -> int:
        return self.config.n_q + 1

    @property
    def num_audio_codebooks(self) -> int:
        return self.config.n_q

    @property
    def audio_offset(self) -> int:
        return 1

    @property
    def initial_token_id(self) -> int:
        return self.config.card

    @property
    def text_initial_token_id(self) -> int:
        return self.config.text_card

    @property
    def text_padding_token_id(self) -> int:
        return self.config.existing_text_padding_id

    @property
    def end_of_text_padding_id(self) -> int:
        # lm.py:260-263: the constructor's `existing_text_end_padding_id` (default 0), carried by LMConfig / loaders
        return self.config.existing_text_end_padding_id

    @property
    def existing_text_padding_id(self) -> int:
        return self.config.existing_text_padding_id

    # lm.py:176-183 builds a ConditionProvider from the checkpoint's `conditioners`; turning attributes (a speaker wav, a text
    # description) into tensors is not on the frame step and is out of scope here.  Callers that branch on
    # `lm.condition_provider is not None` (run_inference.py:38, tts.py:449-460) see None and pass `condition_tensors` to LMGen.
    condition_provider = None

    def set_streaming_detached(self, streaming_detached: bool) -> None:
        """streaming.py:78-86.  The reference detaches a module so that a parent's `.streaming()` does not reach it; an engine
        handle has no parent module and is only ever put into streaming mode by a direct call, i.e. it is always detached."""
        self._streaming_detached = bool(streaming_detached)

    @property
    def zero_token_id(self) -> int:
        return -1

    @property
    def ungenerated_token_id(self) -> int:
        return -2


class LMGen:
    """Streaming generation (reference: lm.py:556-850), including classifier-free guidance (`cfg_coef`,
    `cfg_is_masked_until`, `cfg_is_no_text`), `sum` condition tensors through `lm_model.fuser` and the per-step hooks
    `on_text_logits_hook` / `on_text_hook` / `on_audio_hook` (the step then runs in segments with the callbacks in between,
    mmi_lm_set_hooks), and `cross` condition tensors for models built with cross-attention layers."""

    def __init__(self, lm_model: LMModel, use_sampling: bool = True, temp: float = 0.8, temp_text: float = 0.7,
                 top_k: int = 250, top_k_text: int = 25, cfg_coef: float = 1.0, check: bool = False,
                 condition_tensors=None, on_text_hook=None, on_text_logits_hook=None, on_audio_hook=None,
                 support_out_of_sync: bool = False, cfg_is_masked_until=None, cfg_is_no_text: bool = False,
                 seed: int = 0):
        assert not lm_model.training, "generation shouldn't be used in training mode."
        if cfg_coef != 1.:                                # lm.py:600-603
            if not cfg_is_no_text and not cfg_is_masked_until:
                assert lm_model.fuser is not None, "Model has no fuser, cannot do CFG."
                assert condition_tensors, "Missing condition tensors for CFG."
        self.condition_tensors = condition_tensors
        self.cfg_is_masked_until = cfg_is_masked_until
        self.cfg_is_no_text = cfg_is_no_text
        # per-step hooks (lm.py:568-570, 734-757): each may modify its tensor in place, like the reference's
        self.on_text_hook = on_text_hook
        self.on_text_logits_hook = on_text_logits_hook
        self.on_audio_hook = on_audio_hook
        self._hooks_keep = None
        self._hook_error = None
        self.lm_model = lm_model
        self.use_sampling = use_sampling
        self.temp = temp
        self.temp_text = temp_text
        self.top_k = top_k
        self.top_k_text = top_k_text
        self.cfg_coef = cfg_coef
        self.check = check
        self.support_out_of_sync = support_out_of_sync
        self.max_delay = max(lm_model.delays)
        self.seed = seed
        self._lib = lm_model._lib
        self._batch: Optional[int] = None

    # ---- plumbing ------------------------------------------------------------------------------
    @property
    def device(self) -> torch.device:
        return self.lm_model.device

    def _stream(self):
        return _capi.stream_ptr(self.device)

    def _mask_ptr(self, mask: Optional[torch.Tensor]):
        if mask is None:
            return None, None
        m = mask.to(device=self.device, dtype=torch.bool).contiguous().view(torch.uint8)
        assert m.numel() == self._batch
        return m, m.data_ptr()

    # ---- streaming lifecycle -------------------------------------------------------------------
    @property
    def is_streaming(self) -> bool:
        return self._batch is not None

    def streaming_forever(self, batch_size: int) -> None:
        s = _capi.Sampling()
        s.use_sampling = 1 if self.use_sampling else 0
        s.temp, s.temp_text = self.temp, self.temp_text
        s.top_k, s.top_k_text = self.top_k, self.top_k_text
        s.seed = self.seed
        lm = self.lm_model
        g = _capi.Guidance()
        g.cfg_coef = float(self.cfg_coef)
        g.cfg_is_no_text = 1 if self.cfg_is_no_text else 0
        rows = int(batch_size) * (2 if self.cfg_coef != 1. else 1)
        keep = []
        if lm.fuser is None:                               # lm.py:616-628
            assert not self.condition_tensors
        else:
            assert self.condition_tensors is not None
            cs = lm.fuser.get_sum(self.condition_tensors)
            if cs is not None:
                assert cs.shape[0] == rows, "cfg requires 2x more conditions." if self.cfg_coef != 1. else "one condition row per session"
                cs = cs.to(device=self.device, dtype=torch.bfloat16).contiguous().view(rows, lm.dim)
                keep.append(cs)
                g.condition_sum = cs.data_ptr()
            cx = lm.fuser.get_cross(self.condition_tensors)                     # lm.py:623-627
            if cx is not None:
                assert cx.shape[0] == rows, "cfg requires 2x more conditions." if self.cfg_coef != 1. else "one condition row per session"
                cx = cx.to(device=self.device, dtype=torch.bfloat16).contiguous()
                assert cx.dim() == 3 and cx.shape[2] == lm.dim, cx.shape
                keep.append(cx)
                g.condition_cross = cx.data_ptr()
                g.cross_len = int(cx.shape[1])
        if lm.config.cross_attention:
            assert g.condition_cross, "the model has cross-attention layers: a `cross` condition tensor is required"   # transformer.py:793-795
        if self.cfg_is_masked_until is not None and self.cfg_coef != 1.:
            assert len(self.cfg_is_masked_until) == int(batch_size)
            mu = (C.c_int64 * int(batch_size))(*[int(v) for v in self.cfg_is_masked_until])
            keep.append(mu)
            g.cfg_is_masked_until = C.cast(mu, C.c_void_p)
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        self._lib.check(self._lib.mmi_lm_streaming_start_guided(lm._handle, int(batch_size), C.byref(s), C.byref(g), self._stream()))
        del keep
        self._batch = int(batch_size)
        self._install_hooks()

    def _install_hooks(self) -> None:
        """The reference calls `on_text_logits_hook(text_logits [B,1,1,card])`, `on_text_hook(text_token [B])` and
        `on_audio_hook(audio_tokens [B,dep_q])` between the stages of a step and lets them write in place (lm.py:734-757).
        Here each is a C callback of the segmented step: tensor out (mmi_lm_hook_io), Python hook, tensor back in - all
        stream-ordered, nothing synchronises."""
        lib, h = self._lib, self.lm_model._handle
        if not (self.on_text_logits_hook or self.on_text_hook or self.on_audio_hook):
            lib.check(lib.mmi_lm_set_hooks(h, None))
            self._hooks_keep = None
            return
        cfg, B = self.lm_model.config, self._batch

        def wrap(which, hook, make, view):
            if hook is None:
                return _capi.HOOK_FN()                      # NULL function pointer

            def cb(_user):
                try:
                    t = make()
                    lib.check(lib.mmi_lm_hook_io(h, which, 0, t.data_ptr(), t.numel() * t.element_size(), self._stream()))
                    hook(view(t))
                    lib.check(lib.mmi_lm_hook_io(h, which, 1, t.data_ptr(), t.numel() * t.element_size(), self._stream()))
                    return 0
                except BaseException as e:                  # never unwind through the C frames: report after the step
                    self._hook_error = e
                    return 1
            return _capi.HOOK_FN(cb)
        dev = self.device
        hooks = _capi.LMHooks()
        hooks.on_text_logits = wrap(0, self.on_text_logits_hook,
                                    lambda: torch.empty(B, cfg.text_card, device=dev, dtype=torch.bfloat16),
                                    lambda t: t.view(B, 1, 1, cfg.text_card))
        hooks.on_text_token = wrap(1, self.on_text_hook, lambda: torch.empty(B, device=dev, dtype=torch.int64), lambda t: t)
        hooks.on_audio_tokens = wrap(2, self.on_audio_hook, lambda: torch.empty(B, cfg.dep_q, device=dev, dtype=torch.int64),
                                     lambda t: t)
        self._hooks_keep = hooks                            # the callbacks must outlive the stream
        lib.check(lib.mmi_lm_set_hooks(h, C.byref(hooks)))

    def _stop_streaming(self) -> None:
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        self._lib.mmi_lm_set_hooks(self.lm_model._handle, None)
        self._hooks_keep = None
        self._lib.check(self._lib.mmi_lm_streaming_stop(self.lm_model._handle))
        self._batch = None

    @contextmanager
    def streaming(self, batch_size: int):
        self.streaming_forever(batch_size)
        try:
            yield self
        finally:
            self._stop_streaming()

    def reset_streaming(self, reset_mask: Optional[torch.Tensor] = None) -> None:
        assert self.is_streaming
        keep, ptr = self._mask_ptr(reset_mask)
        self._lib.check(self._lib.mmi_lm_reset(self.lm_model._handle, ptr, self._stream()))

    def get_streaming_state(self) -> dict:
        """streaming.py:158-166: the complete streaming state (a copy: one opaque device tensor + the host step counter)."""
        assert self.is_streaming
        h = self.lm_model._handle
        n = int(self._lib.mmi_lm_state_bytes(h))
        buf = torch.empty(n, dtype=torch.uint8, device=self.device)
        word = C.c_int64(0)
        self._lib.check(self._lib.mmi_lm_state_save(h, buf.data_ptr(), n, C.byref(word), self._stream()))
        return {"lm_gen": buf, "offset_cpu": int(word.value), "batch_size": self._batch}

    def set_streaming_state(self, state: dict) -> None:
        """streaming.py:168-181."""
        assert self.is_streaming
        if "lm_gen" not in state:
            raise RuntimeError("Expected to find a streaming state for lm_gen.")
        buf = state["lm_gen"]
        self._lib.check(self._lib.mmi_lm_state_load(self.lm_model._handle, buf.data_ptr(), buf.numel(), int(state["offset_cpu"]), self._stream()))

    def set_streaming_detached(self, streaming_detached: bool) -> None:
        """streaming.py:78-86 (lm.py:579 calls it on the model): see LMModel.set_streaming_detached."""
        self._streaming_detached = bool(streaming_detached)

    def set_exec_mask(self, exec_mask: torch.Tensor) -> None:
        assert self.is_streaming
        keep, ptr = self._mask_ptr(exec_mask)
        self._lib.check(self._lib.mmi_lm_set_exec_mask(self.lm_model._handle, ptr, self._stream()))

    # ---- test / benchmark aids (no reference counterpart) -----------------------------------------
    def seek(self, offsets) -> None:
        """Move every session to stream position offsets[b] without touching the KV ring (mmi_lm_seek): the skipped positions
        read whatever the ring holds (zeros after `streaming`).  For ring-wrap tests at the real capacity and for timing a
        full-context step without thousands of warm-up steps."""
        assert self.is_streaming
        offs = [int(v) for v in offsets]
        assert len(offs) == self._batch
        arr = (C.c_int64 * len(offs))(*offs)
        self._lib.check(self._lib.mmi_lm_seek(self.lm_model._handle, C.cast(arr, C.c_void_p), self._stream()))

    def hidden_taps(self) -> torch.Tensor:
        """Parity tap (tests): the residual stream of the LAST step after the first and after the last temporal layer, bf16
        [2, model rows, dim].  `lm_model.enable_hidden_taps()` must have been called before `streaming()`."""
        rows = self._lib.mmi_lm_model_rows(self.lm_model._handle)
        out = torch.empty(2, rows, self.lm_model.config.dim, device=self.device, dtype=torch.bfloat16)
        self._lib.check(self._lib.mmi_lm_get_hidden_taps(self.lm_model._handle, out.data_ptr(), out.numel() * 2, self._stream()))
        return out

    def launch_list(self, with_bytes: bool = False):
        """[(site, kernel)] per kernel launch of one step, in launch order (recorded during the first step); with_bytes: a
        third field, the weight bytes a GEMM launch streams."""
        return _capi.launch_list(lambda buf, cap: self._lib.mmi_lm_launch_list(self.lm_model._handle, buf, cap), with_bytes)

    # ---- step ------------------------------------------------------------------------------------
    def _step(self, input_tokens: torch.Tensor, want_taps: bool, noise: Optional[torch.Tensor],
              forced: Optional[torch.Tensor]):
        if not self.is_streaming:
            raise RuntimeError("You should wrap those calls with a `with lm_gen.streaming(): ...`.")
        assert input_tokens.dim() == 3, "Shape should be [B, K, T]."
        B, Ki, S = input_tokens.shape
        assert B == self._batch, f"Got a batch size {B}, expected {self._batch}"
        assert S == 1, "Only support being given steps one by one."
        cfg = self.lm_model.config
        needed = cfg.n_q - cfg.dep_q
        assert Ki >= needed, f"We expect {needed} tokens from the user stream, got {Ki}."
        codes = input_tokens.to(device=self.device, dtype=torch.int64).contiguous()
        if self.check:
            # lm.py:703-711 asserts on the model's input after the delay ring; the only values that enter that ring from outside
            # are these user codes (the rest the engine samples itself), so the same two conditions are asserted on them here -
            # like the reference's, a host synchronisation that only a debugging run pays
            user = codes[:, :needed]
            assert not (user == self.lm_model.ungenerated_token_id).any(), user
            assert (user <= self.lm_model.card).all(), user
        out = torch.empty(B, cfg.dep_q + 1, 1, device=self.device, dtype=torch.int64)
        tl = al = None
        tlp = alp = None
        if want_taps:
            tl = torch.empty(B, cfg.text_card, device=self.device, dtype=torch.float32)
            al = torch.empty(B, cfg.dep_q, cfg.card, device=self.device, dtype=torch.float32)
            tlp, alp = tl.data_ptr(), al.data_ptr()
        npz = None
        if noise is not None:
            kmax = max(self.top_k, self.top_k_text, 1)
            noise = noise.to(device=self.device, dtype=torch.float32).contiguous()
            assert noise.shape == (B, 1 + cfg.dep_q, kmax), f"noise must be [B, 1+dep_q, {kmax}]"
            npz = noise.data_ptr()
        if forced is not None:
            f = forced.to(device=self.device, dtype=torch.int64).contiguous().view(B, 1 + cfg.dep_q)
            self._lib.check(self._lib.mmi_lm_force_next_tokens(self.lm_model._handle, f.data_ptr(), self._stream()))
        valid = C.c_int32(0)
        self._hook_error = None
        rc = self._lib.mmi_lm_step(self.lm_model._handle, codes.data_ptr(), Ki, out.data_ptr(), tlp, alp, npz, B,
                                   C.byref(valid), self._stream())
        if rc and self._hook_error is not None:             # a Python hook raised: surface ITS exception
            err, self._hook_error = self._hook_error, None
            raise err
        self._lib.check(rc)
        if not self.support_out_of_sync and not valid.value:
            return None, tl, al
        return out, tl, al

    @torch.no_grad()
    def step(self, input_tokens: torch.Tensor, depformer_replace_tokens: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """[B, >=8, 1] int64 user codes -> [B, 1 + dep_q, 1] int64 (text + generated audio) or None during the delay."""
        forced = None
        if depformer_replace_tokens is not None:   # lm.py:751-755
            assert depformer_replace_tokens.dim() == 3
            B = depformer_replace_tokens.shape[0]
            forced = torch.full((B, 1 + self.lm_model.dep_q), -1, dtype=torch.int64, device=self.device)
            forced[:, 1:] = depformer_replace_tokens.to(self.device).squeeze(-1)
        out, _, _ = self._step(input_tokens, False, None, forced)
        return out

    @torch.no_grad()
    def step_with_extra_heads(self, input_tokens: torch.Tensor, depformer_replace_tokens: Optional[torch.Tensor] = None):
        """lm.py:793-807: `step` plus softmax(extra_head(transformer_out)) for every extra head, each [model rows, 1, dim]."""
        out = self.step(input_tokens, depformer_replace_tokens)
        if out is None:
            return None
        cfg = self.lm_model.config
        rows = int(self._lib.mmi_lm_model_rows(self.lm_model._handle))
        probs = torch.empty(rows, max(cfg.extra_heads_num_heads, 1), cfg.extra_heads_dim, device=self.device, dtype=torch.float32)
        self._lib.check(self._lib.mmi_lm_extra_heads(self.lm_model._handle, probs.data_ptr(), self._stream()))
        heads = [probs[:, h][:, None].to(torch.bfloat16) for h in range(cfg.extra_heads_num_heads)]
        return out, heads

    @torch.no_grad()
    def step_with_taps(self, input_tokens: torch.Tensor, noise: Optional[torch.Tensor] = None,
                       forced_tokens: Optional[torch.Tensor] = None
                       ) -> Tuple[Optional[torch.Tensor], torch.Tensor, torch.Tensor]:
        """`step` plus the parity taps: text logits [B, text_card] and audio logits [B, dep_q, card] (fp32 copies of
        the bf16 logits the tokens were sampled from); `noise` replaces the RNG, `forced_tokens` teacher-forces."""
        return self._step(input_tokens, True, noise, forced_tokens)
