"""ORACLE (test infrastructure - never imported by the product path).

A numpy restatement of the reference's `LMGen.step` for Moshi (moshi/moshi/models/lm.py:668-850) with the
eager bf16 numerics of the reference PyTorch path: every nn.Linear output, norm output, RoPE output, activation
and residual add is rounded to bf16 (round-to-nearest-even), accumulation and norm/rope/softmax math are fp32.
Used by tests/, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg as the CHECKER of the HIP engine.

Pinned: tests/test_oracle_pinned.py checks it against golden vectors produced by running the reference itself
(tests/golden/make_golden_lm.py).  The reference's scaled-dot-product attention is an ATen backend kernel whose
internal blocking/rounding is not part of the reference sources; here it is restated from its definition (fp32
softmax(QK^T/sqrt(d) + mask) V, output rounded to bf16), which the stated logits tolerance covers.

PARITY UNPINNED for ONE part: the int8 x int8 linear (`QWeight` / `linear_int8`: the reference's `QLinear.forward` calls
bitsandbytes, a dependency that is neither in /root/reference nor installable here).  That part restates the library's published
algorithm and is pinned only on hand-computed known answers; everything else in this file is pinned on the reference's own output.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np

f32 = np.float32


def bf16r(x: np.ndarray) -> np.ndarray:
    """Round fp32 values to the nearest bf16 (ties to even) and return them as fp32."""
    x = np.ascontiguousarray(x, dtype=f32)
    u = x.view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(f32)


class LazyWeight:
    """A linear weight that stays wherever its tensor lives (e.g. bf16 on the GPU) and is widened to fp32 numpy only while a
    `linear` uses it: lets the 32-layer 7B oracle (30 GB in fp32) run on a host that cannot hold it (tests only)."""

    def __init__(self, tensor):
        self.tensor = tensor

    @property
    def shape(self):
        return tuple(self.tensor.shape)

    def numpy(self) -> np.ndarray:
        return self.tensor.detach().cpu().float().numpy()


def _np(t):
    if isinstance(t, LazyWeight):
        return t
    if isinstance(t, np.ndarray):
        return t.astype(f32)
    return t.detach().cpu().float().numpy()


class QWeight:
    """Row-wise int8 weight as `QLinear.__init__` stores it (utils/quantize.py:14-22): `weight` = CB int8 [N, K], `weight_scb` =
    SCB fp32 [N] (the row absmax), W ~= CB * SCB / 127.

    Two forwards (`linear` below):
    * act8 = True - `QLinear.forward` (utils/quantize.py:24-40): `bnb.matmul(x.half(), CB, state)` with a fresh `MatmulLtState`
      (threshold 0: no outlier columns).  bitsandbytes (pinned >= 0.45, < 0.50 by moshi/pyproject.toml:9) is NOT in /root/reference
      and not installed here, so its published algorithm is restated (`int8_vectorwise_quant`, `int8_linear_matmul`,
      `int8_mm_dequant` of bitsandbytes/functional.py, "default" implementations of 0.46+, same arithmetic as the CUDA kernels
      kInt8VectorQuant / kdequant_mm_int32_fp16):
          SCA[b]  = max_k |x[b, k]|                                       (fp32)
          CA[b,k] = int8(round_half_even(x[b, k] * (127 / SCA[b])))       (fp32 product)
          out32   = CA @ CB^T                                             (exact int32)
          y[b,n]  = out32[b, n] * (SCA[b] * SCB[n]) * (1 / (127 * 127))   (fp32)
      PARITY UNPINNED AGAINST bitsandbytes: no reference test exercises QLinear and the library cannot run here; this is a
      restatement of its documented rule, not a check against its output.  One known point where the library's own variants
      differ from each other and from this restatement: the scale 127 / SCA is the correctly rounded fp32 quotient here (and in
      the engine), `tensor.reciprocal() * 127.0` in the library's torch backend (what PyTorch makes of `127.0 / tensor`), the
      approximate `__fdividef` in its CUDA kernel - a code moves by one step where x * scale lands within an ulp of a rounding tie
      (1 in 36 608 codes in tests/test_oracle_pinned.py::test_int8_rule_agrees_with_a_torch_restatement_of_the_library_ops).  Two deliberate differences, both no-ops in range:
      `x.half()` (bf16 -> fp16 is exact for |x| in [2^-14, 65504]; smaller values quantise to 0 either way) is skipped, and
      the result is rounded to the engine's activation type bf16 where bnb returns fp16.
    * act8 = False - weight-only dequantisation (activations stay bf16; the engine's `MMI_Q8_ACT=bf16` mode, rounds 1-3)."""

    def __init__(self, q: np.ndarray, scb: np.ndarray):
        self.q = q.astype(f32)
        self.scb = scb.astype(f32)
        self.scale = (self.scb / f32(127.0)).astype(f32)
        self.act8 = False

    @property
    def shape(self):
        return self.q.shape

    def rows(self, lo: int, hi: int) -> "QWeight":
        w = QWeight.__new__(QWeight)
        w.q, w.scale, w.scb, w.act8 = self.q[lo:hi], self.scale[lo:hi], self.scb[lo:hi], self.act8
        return w


def int8_vectorwise_quant(x: np.ndarray):
    """bitsandbytes.functional.int8_vectorwise_quant (threshold 0) restated: row absmax SCA and CA = round(x * (127 / SCA)),
    half to even, as fp32 integers in [-127, 127]; an all-zero row quantises to zeros."""
    x = np.ascontiguousarray(x, dtype=f32)
    sca = np.abs(x).max(-1, keepdims=True).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        ca = np.rint((x * (f32(127.0) / sca)).astype(f32))
    return np.where(sca > 0, ca, f32(0)).astype(f32), sca


def linear_int8(x: np.ndarray, w: "QWeight") -> np.ndarray:
    """`bnb.matmul` on int8 operands restated (see QWeight): exact integer accumulation (fp32 chunks of <= 1024 products of
    magnitude <= 127 * 127 are exact, summed in fp64), dequantised in fp32, rounded to bf16."""
    ca, sca = int8_vectorwise_quant(x)
    K = ca.shape[-1]
    acc = np.zeros((ca.shape[0], w.q.shape[0]), np.float64)
    for k0 in range(0, K, 1024):
        acc += (ca[:, k0:k0 + 1024] @ w.q[:, k0:k0 + 1024].T).astype(np.float64)
    out = acc.astype(f32) * (sca * w.scb[None, :]).astype(f32) * f32(1.0 / (127.0 * 127.0))
    return bf16r(out.astype(f32))


def e4m3r(x: np.ndarray) -> np.ndarray:
    """Round fp32 values to the nearest OCP e4m3fn value (ties to even), saturating at +-448, returned as fp32: 3 mantissa
    bits for |x| >= 2^-6, a fixed grid of 2^-9 below."""
    x = np.ascontiguousarray(x, dtype=f32)
    a = np.minimum(np.abs(x), f32(448.0))
    m, e = np.frexp(a)                                   # a = m * 2^e, m in [0.5, 1)
    step = np.where(a < f32(2.0 ** -6), f32(2.0 ** -9), np.ldexp(f32(1.0), e - 4)).astype(f32)
    return (np.sign(x) * np.round(a / step) * step).astype(f32)     # np.round: half to even


class F8Weight:
    """fp8 linear of the fp8-MFMA path: W ~= code * weight_scale, activations quantised to e4m3 after dividing by the static
    `input_scale`; products accumulate in fp32 (the MFMA's accumulator)."""

    def __init__(self, codes: np.ndarray, scale: np.ndarray, input_scale: float = 1.0):
        self.q = codes.astype(f32)
        self.scale = scale.astype(f32)
        self.input_scale = f32(input_scale)
        self.noise = None       # (eps, rng): relative perturbation of the accumulator, see LMOracle(fp8_accumulate_noise=...)

    @property
    def shape(self):
        return self.q.shape

    def rows(self, lo: int, hi: int) -> "F8Weight":
        w = F8Weight(self.q[lo:hi], self.scale[lo:hi], self.input_scale)
        w.noise = self.noise
        return w


def linear(x: np.ndarray, w, acc64: bool = False) -> np.ndarray:
    """nn.Linear(bias=False) on bf16 tensors: fp32 accumulate, bf16 result.  acc64: accumulate in fp64 instead (see
    LMOracle(accumulate64=...): the yardstick for how much the summation order of a GEMM moves the network's outputs)."""
    if isinstance(w, F8Weight):
        x8 = e4m3r((x * (f32(1.0) / w.input_scale)).astype(f32))
        acc = (x8 @ w.q.T).astype(f32)
        if w.noise is not None:
            eps, rng = w.noise
            acc = (acc * (1.0 + eps * rng.uniform(-1.0, 1.0, acc.shape))).astype(f32)
        return bf16r((acc * (w.scale * w.input_scale)[None, :]).astype(f32))
    if isinstance(w, QWeight):
        if w.act8:
            return linear_int8(x, w)
        return bf16r(((x @ w.q.T).astype(f32) * w.scale[None, :]).astype(f32))
    if isinstance(w, LazyWeight):
        w = w.numpy()
    if acc64:
        return bf16r((x.astype(np.float64) @ w.T.astype(np.float64)).astype(f32))
    return bf16r((x @ w.T).astype(f32))


def rms_norm(x: np.ndarray, alpha: np.ndarray, eps: float = 1e-8, stat64: bool = False) -> np.ndarray:
    """transformer.py:45-58 with dtype=float32 (`rms_norm_f32`).  stat64: the mean of squares accumulated in fp64 (see
    LMOracle(stat64=...): another correct evaluation of the same function, the yardstick of the int8 network tests)."""
    if stat64:
        var = (np.float64(eps) + np.mean(x.astype(np.float64) ** 2, axis=-1, keepdims=True)).astype(f32)
    else:
        var = f32(eps) + np.mean(x * x, axis=-1, keepdims=True, dtype=f32)
    return bf16r(x * (alpha.reshape(1, -1) * (f32(1.0) / np.sqrt(var))).astype(f32))


def layer_norm(x: np.ndarray, w: np.ndarray, b: np.ndarray, eps: float = 1e-5) -> np.ndarray:
    """nn.LayerNorm on a bf16 tensor (the layer's `norm_cross`, transformer.py:731-732): statistics and affine in fp32, one
    rounding to bf16 at the end."""
    mean = np.mean(x, axis=-1, keepdims=True, dtype=f32)
    var = np.mean((x - mean) * (x - mean), axis=-1, keepdims=True, dtype=f32)
    y = (x - mean) * (f32(1.0) / np.sqrt(var + f32(eps))).astype(f32)
    return bf16r((y * w.reshape(1, -1) + b.reshape(1, -1)).astype(f32))


def silu(x: np.ndarray) -> np.ndarray:
    return (x / (1.0 + np.exp(-x))).astype(x.dtype)


def gated_ffn(x: np.ndarray, w_in: np.ndarray, w_out: np.ndarray, acc64: bool = False, stat64: bool = False) -> np.ndarray:
    """gating.py:13-22 in eager bf16: linear_in -> silu(gate) * value -> linear_out."""
    h = linear(x, w_in, acc64)
    H = h.shape[-1] // 2
    act = bf16r(silu(h[:, :H].astype(np.float64)).astype(f32)) if stat64 else bf16r(silu(h[:, :H]))
    return linear(bf16r(act * h[:, H:]), w_out, acc64)


def rope_1(q: np.ndarray, k: np.ndarray, offset: np.ndarray, max_period: float):
    """rope.py:11-82 for T=1, layout [B, H, D], interleaved; fp32 math, bf16 result."""
    B, H, D = q.shape
    ds = np.arange(D // 2, dtype=f32)
    freqs = np.exp(ds * f32(-math.log(max_period) * 2 / D)).astype(f32)
    ang = (freqs[None, :] * offset.astype(f32)[:, None]).astype(f32)[:, None, :]
    c, s = np.cos(ang).astype(f32), np.sin(ang).astype(f32)

    def rot(x):
        x = x.reshape(B, H, D // 2, 2)
        xr, xi = x[..., 0], x[..., 1]
        return bf16r(np.stack([xr * c - xi * s, xr * s + xi * c], -1).reshape(B, H, D))

    return rot(q), rot(k)


def sample_token(logits: np.ndarray, use_sampling: bool, temp: float, top_k: int, noise: Optional[np.ndarray]) -> np.ndarray:
    """sampling.py:86-106 on fp32 logits [B, V].  `noise` [B, k]: the Exp(1) draws of `multinomial` (:40-47), indexed by
    rank in the descending top-k (torch.topk order; ties by index)."""
    B, V = logits.shape
    if not use_sampling or temp <= 0:
        return logits.argmax(-1)
    x = (logits / f32(temp)).astype(f32)
    x = x - x.max(-1, keepdims=True)
    e = np.exp(x).astype(f32)
    p = (e / e.sum(-1, keepdims=True, dtype=f32)).astype(f32)
    k = min(top_k, V)
    out = np.zeros(B, np.int64)
    for b in range(B):
        order = np.lexsort((np.arange(V), -p[b]))[:k]          # value descending, index ascending
        score = p[b][order] / noise[b, :k]
        out[b] = order[int(score.argmax())]
    return out


class LMOracle:
    def __init__(self, state_dict, cfg, fp8_accumulate_noise: float = 0.0, noise_seed: int = 0, accumulate64: bool = False,
                 int8_activations: bool = True, stat64: bool = False):
        """fp8_accumulate_noise: relative perturbation applied to every fp8 GEMM accumulator.  The gfx950 fp8 dot-product unit
        does not sum its 8-product groups exactly: products below ~2^-13 of the group's largest are shifted out (measured by
        scripts/fp8_probe.hip: up to 2.7e-4 of sum|products|).  The tests use this knob to measure how far such a perturbation
        moves the fp8 network's own outputs (e4m3 re-quantisation of every activation amplifies it), which is the yardstick an
        fp8 implementation on that hardware can be held to."""
        # accumulate64: every bf16 GEMM accumulates in fp64 instead of fp32 - the same network with a different (here: exact)
        # summation; its distance from the fp32-accumulating oracle is how far ANY correct implementation may sit from either
        # (bf16 rounding flips of a few GEMM outputs, amplified layer by layer): the yardstick of the full-depth parity test
        self.acc64 = bool(accumulate64)
        # stat64: the transcendental / reduction parts that are NOT a linear - RMSNorm's mean of squares, the softmax of both
        # attentions, the SiLU - evaluated in fp64 and rounded once.  Under int8 x int8 linears (exact integer GEMMs, identical in
        # every correct implementation) these are the only places where two correct implementations can differ by a rounding flip;
        # the distance of this variant from the plain oracle is the yardstick the int8 network tests hold the engine to
        self.stat64 = bool(stat64)
        self.cfg = cfg
        sd = {k: _np(v) for k, v in state_dict.items()}
        # int8_activations: the int8 linears run bitsandbytes' int8 x int8 rule (the reference's QLinear.forward; default) or
        # weight-only dequantisation (False).  Cross-attention linears of a conditioned model stay weight-only in the engine.
        for k in [k for k in sd if k.endswith("_scb")]:          # int8 linears: `weight` (int8) + `weight_scb`
            sd[k[:-4]] = QWeight(sd[k[:-4]], sd.pop(k))
            sd[k[:-4]].act8 = bool(int8_activations) and ".cross_attention." not in k
        for k in [k for k in sd if k.endswith(".weight_scale")]:  # fp8 linears: `weight` (e4m3fn) + `weight_scale` [+ `input_scale`]
            stem = k[: -len(".weight_scale")]
            ins = sd.pop(stem + ".input_scale", None)
            sd[stem + ".weight"] = F8Weight(sd[stem + ".weight"], sd.pop(k), 1.0 if ins is None else float(np.asarray(ins).reshape(-1)[0]))
            if fp8_accumulate_noise > 0:
                if not hasattr(self, "_noise_rng"):
                    self._noise_rng = np.random.default_rng(noise_seed)
                sd[stem + ".weight"].noise = (float(fp8_accumulate_noise), self._noise_rng)
        c = cfg
        self.emb = [sd[f"emb.{i}.weight"] for i in range(c.n_q)]
        self.text_emb = sd["text_emb.weight"]
        self.text_linear = sd["text_linear.weight"]
        self.out_norm = sd["out_norm.alpha"].reshape(-1)
        self.layers = []
        for l in range(c.num_layers):
            p = f"transformer.layers.{l}"
            self.layers.append(dict(in_proj=sd[p + ".self_attn.in_projs.0.weight"], out_proj=sd[p + ".self_attn.out_projs.0.weight"],
                                    n1=sd[p + ".norm1.alpha"].reshape(-1), n2=sd[p + ".norm2.alpha"].reshape(-1),
                                    w_in=sd[p + ".gating.linear_in.weight"], w_out=sd[p + ".gating.linear_out.weight"]))
            if getattr(c, "cross_attention", False):             # transformer.py:727-732
                self.layers[-1].update(x_in=sd[p + ".cross_attention.in_projs.0.weight"], x_out=sd[p + ".cross_attention.out_projs.0.weight"],
                                       nx_w=sd[p + ".norm_cross.weight"].reshape(-1), nx_b=sd[p + ".norm_cross.bias"].reshape(-1))
        self.dep_in = [sd[f"depformer_in.{k}.weight"] for k in range(c.dep_q)]
        self.dep_emb = ([sd["depformer_text_emb.weight"]] + [sd[f"depformer_emb.{k}.weight"] for k in range(c.dep_q - 1)]) if c.dep_q > 0 else []
        self.dep_layers = []
        for l in range(c.depformer_num_layers if c.dep_q > 0 else 0):      # dep_q == 0: an ASR model without depformer (lm.py:218-221)
            p = f"depformer.layers.{l}"
            self.dep_layers.append(dict(
                in_proj=[sd[p + f".self_attn.in_projs.{k}.weight"] for k in range(c.dep_q)],
                out_proj=[sd[p + f".self_attn.out_projs.{k}.weight"] for k in range(c.dep_q)],
                n1=sd[p + ".norm1.alpha"].reshape(-1), n2=sd[p + ".norm2.alpha"].reshape(-1),
                w_in=[sd[p + f".gating.{k}.linear_in.weight"] for k in range(c.dep_q)],
                w_out=[sd[p + f".gating.{k}.linear_out.weight"] for k in range(c.dep_q)]))
        self.linears = [sd[f"linears.{k}.weight"] for k in range(c.dep_q)]
        self.extra_heads = []                                   # lm.py:224-226
        while f"extra_heads.{len(self.extra_heads)}.weight" in sd:
            self.extra_heads.append(sd[f"extra_heads.{len(self.extra_heads)}.weight"])
        self.B = 0

    # ---- streaming state (lm.py:605-666; transformer.py:448-486) -----------------------------------
    def streaming(self, B: int, cfg_coef: float = 1.0, cfg_is_no_text: bool = False, cfg_is_masked_until=None,
                  condition_sum=None, condition_cross=None):
        """`condition_sum`: the fuser's summed condition [model rows, dim] (lm.py:621-628); with guidance (cfg_coef != 1) the
        model runs 2B rows, conditioned half first (lm.py:646-651)."""
        c = self.cfg
        self.cfg_coef, self.cfg_is_no_text = float(cfg_coef), bool(cfg_is_no_text)
        self.cfg_is_masked_until = None if cfg_is_masked_until is None else np.asarray(cfg_is_masked_until, np.int64)
        self.gen_B = B
        self.condition_sum = None if condition_sum is None else bf16r(_np(condition_sum).reshape(-1, c.dim))
        if self.cfg_coef != 1.0:
            B = 2 * B
        if self.condition_sum is not None:
            assert self.condition_sum.shape[0] == B, "cfg requires 2x more conditions."
        # `condition_cross`: the fuser's cross-attention source [model rows, T_c, dim] (lm.py:623-627); every temporal layer
        # projects it ONCE to keys / values (transformer.py:495-531: cached on the first step), bf16 like any nn.Linear output
        self.cross_kv = None
        if getattr(c, "cross_attention", False):
            assert condition_cross is not None, "the model has cross-attention layers: a `cross` condition is required"
            src = bf16r(_np(condition_cross))
            assert src.shape[0] == B and src.shape[2] == c.dim, "cfg requires 2x more conditions."
            H, Dh = c.num_heads, c.dim // c.num_heads
            self.cross_kv = []
            for L in self.layers:
                w = L["x_in"]
                kv = linear(src.reshape(-1, c.dim), w[c.dim:], self.acc64).reshape(B, src.shape[1], 2, H, Dh)
                self.cross_kv.append((kv[:, :, 0].transpose(0, 2, 1, 3), kv[:, :, 1].transpose(0, 2, 1, 3)))   # [B, H, T_c, Dh]
        self._streaming_model(B)
        B = self.gen_B
        self.B = B
        self.exec_mask = np.ones(B, bool)
        self.CT = max(c.delays) + 2
        self.cache = np.full((B, c.n_q + 1, self.CT), -2, np.int64)
        self.offsets = np.zeros(B, np.int64)
        self.offset_cpu = 0

    def _model_rows(self, per_session: np.ndarray) -> np.ndarray:
        """Per-session quantity -> per model row (`.repeat(2)` under guidance, lm.py:655-663)."""
        return np.concatenate([per_session, per_session]) if self.cfg_coef != 1.0 else per_session

    def _guide(self, logits: np.ndarray) -> np.ndarray:
        """logits_null + (logits - logits_null) * cfg_coef on bf16 tensors (lm.py:733, 830-832)."""
        B = self.gen_B
        lg, null = logits[:B], logits[B:]
        return bf16r(null + bf16r(bf16r(lg - null) * f32(self.cfg_coef)))

    def _streaming_model(self, B: int):
        c = self.cfg
        self.model_B = B
        self.CT = max(c.delays) + 2
        H, Dh = c.num_heads, c.dim // c.num_heads
        self.kv = [np.zeros((2, B, H, c.context, Dh), f32) for _ in range(c.num_layers)]
        self.hidden_taps = {}      # {0: x after the first temporal layer, 1: after the last} of the last forward_text ([rows, dim], bf16 values)
        self.tr_offset = np.zeros(B, np.int64)          # MHA offset == RingKVCache.end_offset for every layer

    def reset_streaming(self, mask=None):
        mask = np.ones(self.B, bool) if mask is None else np.asarray(mask, bool)
        self.exec_mask[mask] = True
        self.offsets[mask] = 0
        self.tr_offset[self._model_rows(mask)] = 0
        self.offset_cpu = 0

    def set_exec_mask(self, mask):
        self.exec_mask = np.asarray(mask, bool).copy()

    def seek(self, offsets):
        """The checker's side of `mmi_lm_seek`: every session jumps to stream position offsets[b]; the ring keeps its contents."""
        offsets = np.asarray(offsets, np.int64)
        self.offsets = offsets.copy()
        self.tr_offset = self._model_rows(offsets).copy()
        self.offset_cpu = int(offsets.max())

    # ---- temporal transformer (lm.py:379-408; transformer.py:533-597, 752-802) -------------------------
    def _embed(self, emb: np.ndarray, tok: np.ndarray) -> np.ndarray:
        """ScaledEmbedding.forward (lm_utils.py:102-124): clamp(min=0) gather, zero vector for token -1."""
        y = emb[np.maximum(tok, 0)]
        return np.where((tok == -1)[:, None], f32(0), y).astype(f32)

    def forward_text(self, tokens: np.ndarray):
        c = self.cfg
        B = tokens.shape[0]
        x = None
        for i in range(c.n_q):
            e = self._embed(self.emb[i], tokens[:, i + 1])
            x = e if x is None else bf16r(x + e)
        x = bf16r(x + self._embed(self.text_emb, tokens[:, 0]))
        if self.condition_sum is not None:                       # lm.py:399-400
            x = bf16r(x + self.condition_sum)
        H, Dh, cap = c.num_heads, c.dim // c.num_heads, c.context
        off = self.tr_offset
        exec_rows = self._model_rows(self.exec_mask)
        for l, L in enumerate(self.layers):
            qkv = linear(rms_norm(x, L["n1"], stat64=self.stat64), L["in_proj"], self.acc64).reshape(B, 3, H, Dh)
            q, k = rope_1(qkv[:, 0], qkv[:, 1], off, c.max_period)
            v = qkv[:, 2]
            cache = self.kv[l]
            att = np.zeros((B, H, Dh), f32)
            for b in range(B):
                slot = int(off[b] % cap)
                kq, vq = (e4m3r(k[b]), e4m3r(v[b])) if getattr(c, "kv_cache_dtype", "bf16") == "fp8" else (k[b], v[b])
                cache[0, b, :, slot] = kq                # written unconditionally (transformer.py:243-250)
                cache[1, b, :, slot] = vq
                last = int(off[b])
                end_new = last + 1 if exec_rows[b] else last
                idx = np.arange(cap)
                delta = idx - (last % cap)
                pos = np.where(delta <= 0, last + delta, last + delta - cap)
                pos = np.where(idx >= end_new, -1, pos)
                # non-executing rows: the reference masks with the un-advanced end_offset; their output is don't-care
                dq = last - pos
                ok = (pos >= 0) & (dq >= 0) & (dq < c.context)
                if not ok.any():
                    continue
                if self.stat64:
                    s = (cache[0, b][:, ok].astype(np.float64) @ q[b][:, :, None].astype(np.float64))[..., 0] / math.sqrt(Dh)
                    p = np.exp(s - s.max(-1, keepdims=True))
                    p = p / p.sum(-1, keepdims=True)
                    att[b] = np.einsum("hn,hnd->hd", p, cache[1, b][:, ok].astype(np.float64)).astype(f32)
                    continue
                s = (cache[0, b][:, ok] @ q[b][:, :, None])[..., 0] / f32(math.sqrt(Dh))   # [H, n]
                s = s - s.max(-1, keepdims=True)
                p = np.exp(s).astype(f32)
                p = p / p.sum(-1, keepdims=True, dtype=f32)
                att[b] = np.einsum("hn,hnd->hd", p, cache[1, b][:, ok]).astype(f32)
            att = bf16r(att.reshape(B, H * Dh))
            x = bf16r(x + linear(att, L["out_proj"], self.acc64))
            if self.cross_kv is not None:                        # _cross_attention_block (transformer.py:779-786): no mask, no rope
                kc, vc = self.cross_kv[l]
                q = linear(layer_norm(x, L["nx_w"], L["nx_b"]), L["x_in"][:c.dim], self.acc64).reshape(B, H, Dh)
                s = np.einsum("bhd,bhtd->bht", q, kc).astype(f32) / f32(math.sqrt(Dh))
                s = s - s.max(-1, keepdims=True)
                p = np.exp(s).astype(f32)
                p = p / p.sum(-1, keepdims=True, dtype=f32)
                xa = bf16r(np.einsum("bht,bhtd->bhd", p, vc).astype(f32).reshape(B, H * Dh))
                x = bf16r(x + linear(xa, L["x_out"], self.acc64))
            x = bf16r(x + gated_ffn(rms_norm(x, L["n2"], stat64=self.stat64), L["w_in"], L["w_out"], self.acc64, self.stat64))
            if l == 0 or l == len(self.layers) - 1:              # what a forward hook on layers[0] / layers[-1] returns
                self.hidden_taps[0 if l == 0 else 1] = x.copy()
                if len(self.layers) == 1:
                    self.hidden_taps[1] = x.copy()
        tout = rms_norm(x, self.out_norm, stat64=self.stat64)
        return tout, linear(tout, self.text_linear, self.acc64)

    # ---- depformer (lm.py:450-493, 809-850) ------------------------------------------------------------
    def depformer_step(self, text_token, tout, use_sampling, temp, top_k, noise, forced):
        c = self.cfg
        B = tout.shape[0]                                        # model rows (2x the sessions under guidance)
        Hd, Dhd = c.depformer_num_heads, c.depformer_dim // c.depformer_num_heads
        prev = self._model_rows(text_token)
        keys: List[List[np.ndarray]] = [[] for _ in self.dep_layers]
        vals: List[List[np.ndarray]] = [[] for _ in self.dep_layers]
        tokens, logits_all = [], []
        for k in range(c.dep_q):
            x = bf16r(linear(tout, self.dep_in[k], self.acc64) + self._embed(self.dep_emb[k], prev))
            for l, L in enumerate(self.dep_layers):
                qkv = linear(rms_norm(x, L["n1"], stat64=self.stat64), L["in_proj"][k], self.acc64).reshape(B, 3, Hd, Dhd)
                keys[l].append(qkv[:, 1]); vals[l].append(qkv[:, 2])
                K = np.stack(keys[l], 2); V = np.stack(vals[l], 2)                       # [B, Hd, k+1, Dhd]
                if self.stat64:
                    s = np.einsum("bhd,bhnd->bhn", qkv[:, 0].astype(np.float64), K.astype(np.float64)) / math.sqrt(Dhd)
                    p = np.exp(s - s.max(-1, keepdims=True))
                    p = p / p.sum(-1, keepdims=True)
                    att = bf16r(np.einsum("bhn,bhnd->bhd", p, V.astype(np.float64)).astype(f32).reshape(B, Hd * Dhd))
                else:
                    s = np.einsum("bhd,bhnd->bhn", qkv[:, 0], K).astype(f32) / f32(math.sqrt(Dhd))
                    s = s - s.max(-1, keepdims=True)
                    p = np.exp(s).astype(f32)
                    p = p / p.sum(-1, keepdims=True, dtype=f32)
                    att = bf16r(np.einsum("bhn,bhnd->bhd", p, V).astype(f32).reshape(B, Hd * Dhd))
                x = bf16r(x + linear(att, L["out_proj"][k], self.acc64))
                x = bf16r(x + gated_ffn(rms_norm(x, L["n2"], stat64=self.stat64), L["w_in"][k], L["w_out"][k], self.acc64, self.stat64))
            lg = linear(x, self.linears[k], self.acc64)
            if self.cfg_coef != 1.0:
                lg = self._guide(lg)
            nz = None if noise is None else noise[:, 1 + k]
            tok = sample_token(lg, use_sampling, temp, top_k, nz)
            if forced is not None:
                tok = np.where(forced[:, 1 + k] >= 0, forced[:, 1 + k], tok)
            tokens.append(tok); logits_all.append(lg)
            prev = self._model_rows(tok)
        if not tokens:
            return np.zeros((self.gen_B, 0), np.int64), np.zeros((self.gen_B, 0, c.card), f32)
        return np.stack(tokens, 1), np.stack(logits_all, 1)

    # ---- LMGen._step (lm.py:668-783) ---------------------------------------------
    def step(self, user_codes: np.ndarray, use_sampling: bool = False, temp: float = 0.8, temp_text: float = 0.7,
             top_k: int = 250, top_k_text: int = 25, noise: Optional[np.ndarray] = None,
             forced: Optional[np.ndarray] = None, support_out_of_sync: bool = False):
        c = self.cfg
        B = self.B
        delays = np.asarray(c.delays)
        need = c.n_q - c.dep_q
        codes = np.asarray(user_codes)[:, :need, 0]
        CT = self.CT
        ex = self.exec_mask
        for j in range(need):
            ch = c.dep_q + 1 + j
            wp = (self.offsets + delays[ch]) % CT
            for b in range(B):
                if ex[b]:
                    self.cache[b, ch, wp[b]] = codes[b, j]
        is_init = (self.offsets[:, None] <= delays[None, :]) | ~ex[:, None]
        inp = self.cache[np.arange(B)[:, None], np.arange(c.n_q + 1)[None, :], (self.offsets % CT)[:, None]]
        initial = np.array([c.text_card] + [c.card] * c.n_q)
        inp = np.where(is_init, initial[None, :], inp)
        if self.cfg_coef != 1.0:                                  # lm.py:712-725
            null = inp.copy()
            if self.cfg_is_masked_until is not None:
                zeroed = self.offsets[:, None] <= delays[None, :] + self.cfg_is_masked_until[:, None]
                null = np.where(zeroed & ~is_init, -1, null)
            if self.cfg_is_no_text:
                null[:, 0] = np.where(~is_init[:, 0], -1, null[:, 0])
            inp = np.concatenate([inp, null], 0)
        tout, text_logits = self.forward_text(inp)
        self.last_tout = tout
        if self.cfg_coef != 1.0:                                  # lm.py:727-733
            text_logits = text_logits[:B] if self.cfg_is_no_text else self._guide(text_logits)
        exm = self._model_rows(ex)
        self.tr_offset = np.where(exm, self.tr_offset + 1, self.tr_offset)
        text_token = sample_token(text_logits, use_sampling, temp_text, top_k_text, None if noise is None else noise[:, 0])
        if forced is not None:
            text_token = np.where(forced[:, 0] >= 0, forced[:, 0], text_token)
        audio_tokens, audio_logits = self.depformer_step(text_token, tout, use_sampling, temp, top_k, noise, forced)
        self.offsets = np.where(ex, self.offsets + 1, self.offsets)
        self.offset_cpu += 1
        pos = self.offsets % CT
        for b in range(B):
            if ex[b]:
                self.cache[b, 0, pos[b]] = text_token[b]
                self.cache[b, 1:c.dep_q + 1, pos[b]] = audio_tokens[b]
        taps = (text_logits, audio_logits, text_token, audio_tokens)
        max_delay = int(delays.max())
        if not support_out_of_sync and self.offset_cpu <= max_delay:
            return None, taps
        gd = delays[:c.dep_q + 1]
        index = (self.offsets[:, None] - max_delay + gd[None, :]) % CT
        out = self.cache[np.arange(B)[:, None], np.arange(c.dep_q + 1)[None, :], index]
        hide = (self.offsets <= max_delay) | ~ex
        out = np.where(hide[:, None], -2, out)
        return out[:, :, None], taps

    def extra_head_probs(self) -> np.ndarray:
        """`step_with_extra_heads` (lm.py:793-807): softmax(extra_head(transformer_out)) per head, on the last step's
        transformer output -> [model rows, n_heads, extra_heads_dim] (bf16 values)."""
        outs = []
        for w in self.extra_heads:
            lg = linear(self.last_tout, w)
            e = np.exp(lg - lg.max(-1, keepdims=True)).astype(f32)
            outs.append(bf16r(e / e.sum(-1, keepdims=True, dtype=f32)))
        return np.stack(outs, 1)
