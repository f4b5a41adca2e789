"""ORACLE (test infrastructure - never imported by the product path).

A numpy fp32 restatement of the reference's streaming Mimi codec, used by tests/, by
`__graft_entry__.smoke()` and by bench.py's `cpu_baseline` leg as the CHECKER of the HIP engine.
Every function cites the reference lines (paths relative to the reference checkout) it follows.
It keeps the reference's own state representation (`previous`, `partial`, ring KV + `end_offset`), not
the engine's, so that the two are independent.

Pinned: tests/test_oracle_pinned.py checks this file against golden vectors produced by running the
reference itself (tests/golden/make_golden.py imports /root/reference in the build container).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np

try:  # exact erf for the GELU; scipy is part of the image
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf, otypes=[np.float32])

f32 = np.float32


def _np(t) -> np.ndarray:
    if isinstance(t, np.ndarray):
        return t
    return t.detach().cpu().float().numpy()


def elu(x: np.ndarray) -> np.ndarray:
    """nn.ELU(alpha=1) (seanet.py:63,205,222)."""
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0))).astype(f32)


def gelu(x: np.ndarray) -> np.ndarray:
    """F.gelu, exact erf form (transformer.py:648)."""
    return (f32(0.5) * x * (f32(1.0) + _erf(x * f32(0.70710678118654752440)).astype(f32))).astype(f32)


class StreamingConv1d:
    """conv.py:172-274 (`previous`/`first` state :161-169,233-243; forward :245-274)."""

    def __init__(self, weight, bias, stride: int, pad_mode: str = "constant"):
        self.w = _np(weight).astype(f32)            # [Cout, Cin, K]
        self.b = None if bias is None else _np(bias).astype(f32)
        self.S = stride
        self.K = self.w.shape[2]
        self.pad_mode = pad_mode
        self.previous: Optional[np.ndarray] = None
        self.first: Optional[np.ndarray] = None

    def init_state(self, B: int):
        self.previous = np.zeros((B, self.w.shape[1], self.K - self.S), f32)
        self.first = np.ones(B, bool)

    def reset(self, mask: np.ndarray):
        self.previous[mask] = 0
        self.first[mask] = True

    def __call__(self, x: np.ndarray, exec_mask: np.ndarray) -> np.ndarray:
        B, C, T = x.shape
        S, K = self.S, self.K
        assert T > 0 and T % S == 0, "Steps must be multiple of stride"
        TP = K - S
        if TP and self.pad_mode == "replicate":
            sel = self.first & exec_mask
            self.previous[sel] = x[sel][..., :1]
        xc = np.concatenate([self.previous, x], axis=-1) if TP else x
        T_out = (xc.shape[-1] - K) // S + 1
        idx = (np.arange(T_out)[:, None] * S + np.arange(K)[None, :])          # [T_out, K]
        win = xc[:, :, idx]                                                     # [B, Cin, T_out, K]
        y = np.einsum("bitk,oik->bot", win, self.w, optimize=True).astype(f32)
        if self.b is not None:
            y = y + self.b[None, :, None]
        if TP:
            self.previous[exec_mask] = xc[exec_mask][..., -TP:]
            if self.pad_mode == "replicate":
                self.first[exec_mask] = False
        return y.astype(f32)


class StreamingConvTranspose1d:
    """conv.py:289-362 (`partial` state :277-286,330-338; forward :340-362). groups = 1 or depthwise."""

    def __init__(self, weight, bias, stride: int, depthwise: bool = False):
        self.w = _np(weight).astype(f32)            # [Cin, Cout/groups, K]
        self.b = None if bias is None else _np(bias).astype(f32)
        self.S = stride
        self.K = self.w.shape[2]
        self.depthwise = depthwise
        self.cout = self.w.shape[0] if depthwise else self.w.shape[1]
        self.partial: Optional[np.ndarray] = None

    def init_state(self, B: int):
        self.partial = np.zeros((B, self.cout, self.K - self.S), f32)

    def reset(self, mask: np.ndarray):
        self.partial[mask] = 0

    def __call__(self, x: np.ndarray, exec_mask: np.ndarray) -> np.ndarray:
        B, C, T = x.shape
        S, K = self.S, self.K
        if self.depthwise:
            contrib = x[:, :, :, None] * self.w[None, :, 0, None, :]            # [B, C, T, K]
        else:
            contrib = np.einsum("bit,iok->botk", x, self.w, optimize=True)       # [B, Cout, T, K]
        y = np.zeros((B, self.cout, (T - 1) * S + K), f32)
        for t in range(T):
            y[:, :, t * S:t * S + K] += contrib[:, :, t, :]
        if self.b is not None:
            y = y + self.b[None, :, None]
        PT = K - S
        if PT > 0:
            y[..., :PT] += self.partial
            for_partial = y[..., -PT:].copy()
            if self.b is not None:
                for_partial -= self.b[None, :, None]
            self.partial[exec_mask] = for_partial[exec_mask]
            y = y[..., :-PT]
        return y.astype(f32)


class RingKV:
    """transformer.py:196-288 (RingKVCache.complete)."""

    def __init__(self, B: int, H: int, D: int, cap: int):
        self.cap = cap
        self.cache = np.zeros((2, B, H, cap, D), f32)
        self.end_offset = np.zeros(B, np.int64)

    def complete(self, k: np.ndarray, v: np.ndarray, exec_mask: np.ndarray):
        B, H, T, D = k.shape
        cap = self.cap
        for b in range(B):
            idx = (self.end_offset[b] + np.arange(T)) % cap
            self.cache[0, b][:, idx] = k[b]
            self.cache[1, b][:, idx] = v[b]
        indexes = np.arange(cap)[None, :]
        last = (self.end_offset + T - 1)[:, None]
        end_index = last % cap
        delta = indexes - end_index
        positions = np.where(delta <= 0, last + delta, last + delta - cap)
        self.end_offset = np.where(exec_mask, self.end_offset + T, self.end_offset)
        invalid = indexes >= self.end_offset[:, None]
        positions = np.where(invalid, -1, positions)
        return self.cache[0], self.cache[1], positions


def apply_rope(q: np.ndarray, k: np.ndarray, offset: np.ndarray, max_period: float):
    """rope.py:11-82, interleaved, `time_before_heads=False` layout [B, H, T, D]; all math in fp32."""
    B, H, T, D = q.shape
    ds = np.arange(D // 2, dtype=f32)
    freqs = np.exp(ds * f32(-math.log(max_period) * 2 / D)).astype(f32)
    ts = (offset.astype(f32)[:, None] + np.arange(T, dtype=f32)[None, :]).reshape(B, 1, T, 1)
    ang = (freqs * ts).astype(f32)
    rotr, roti = np.cos(ang).astype(f32), np.sin(ang).astype(f32)

    def rot(x):
        x = x.reshape(B, H, T, D // 2, 2)
        xr, xi = x[..., 0], x[..., 1]
        outr = xr * rotr - xi * roti
        outi = xr * roti + xi * rotr
        return np.stack([outr, outi], axis=-1).reshape(B, H, T, D).astype(f32)

    return rot(q), rot(k)


def layer_norm(x: np.ndarray, w: np.ndarray, b: np.ndarray, eps: float = 1e-5) -> np.ndarray:
    """nn.LayerNorm over the last axis (transformer.py:125-126)."""
    mean = x.mean(-1, keepdims=True, dtype=f32)
    var = ((x - mean) ** 2).mean(-1, keepdims=True, dtype=f32)
    return ((x - mean) / np.sqrt(var + f32(eps)) * w + b).astype(f32)


class TransformerLayer:
    """transformer.py:609-802 with the Mimi options (layer_norm, plain GELU FFN, LayerScale, RoPE, causal+context)."""

    def __init__(self, sd: Dict[str, np.ndarray], prefix: str, H: int, context: int, max_period: float):
        g = lambda n: _np(sd[prefix + n]).astype(f32)
        self.in_proj = g(".self_attn.in_projs.0.weight")
        self.out_proj = g(".self_attn.out_projs.0.weight")
        self.n1w, self.n1b = g(".norm1.weight"), g(".norm1.bias")
        self.n2w, self.n2b = g(".norm2.weight"), g(".norm2.bias")
        self.l1, self.l2 = g(".linear1.weight"), g(".linear2.weight")
        self.ls1, self.ls2 = g(".layer_scale_1.scale"), g(".layer_scale_2.scale")
        self.H, self.context, self.max_period = H, context, max_period
        self.kv: Optional[RingKV] = None
        self.offset: Optional[np.ndarray] = None

    def init_state(self, B: int):
        d = self.out_proj.shape[0]
        self.kv = RingKV(B, self.H, d // self.H, self.context)
        self.offset = np.zeros(B, np.int64)

    def reset(self, mask: np.ndarray):
        self.offset[mask] = 0
        self.kv.end_offset[mask] = 0

    def attention(self, x: np.ndarray, exec_mask: np.ndarray) -> np.ndarray:
        """transformer.py:533-597."""
        B, T, d = x.shape
        H, D = self.H, d // self.H
        proj = (x @ self.in_proj.T).astype(f32)
        q, k, v = [proj[:, :, i * d:(i + 1) * d].reshape(B, T, H, D).transpose(0, 2, 1, 3) for i in range(3)]
        q, k = apply_rope(q, k, self.offset, self.max_period)
        keys, values, pos_k = self.kv.complete(k, v, exec_mask)
        pos_q = self.offset[:, None, None] + np.arange(T)[None, :, None]
        delta = pos_q - pos_k[:, None, :]
        bias = (pos_k[:, None, :] >= 0) & (delta >= 0) & (delta < self.context)        # [B, T, cap]
        att = np.einsum("bhtd,bhsd->bhts", q, keys, optimize=True).astype(f32) / f32(math.sqrt(D))
        att = np.where(bias[:, None], att, -np.inf)
        with np.errstate(invalid="ignore"):   # a row that never executed has no valid key: NaN, as in the reference
            att = att - att.max(-1, keepdims=True)
            p = np.exp(att).astype(f32)
            p = (p / p.sum(-1, keepdims=True, dtype=f32)).astype(f32)
        o = np.einsum("bhts,bhsd->bhtd", p, values, optimize=True).astype(f32)
        o = o.transpose(0, 2, 1, 3).reshape(B, T, d)
        self.offset = np.where(exec_mask, self.offset + T, self.offset)
        return (o @ self.out_proj.T).astype(f32)

    def __call__(self, x: np.ndarray, exec_mask: np.ndarray) -> np.ndarray:
        x = x + self.ls1 * self.attention(layer_norm(x, self.n1w, self.n1b), exec_mask)
        h = gelu((layer_norm(x, self.n2w, self.n2b) @ self.l1.T).astype(f32))
        x = x + self.ls2 * (h @ self.l2.T).astype(f32)
        return x.astype(f32)


def cdist_argmin(x: np.ndarray, E: np.ndarray) -> np.ndarray:
    """EuclideanCodebook._quantize (core_vq.py:270-276): torch.cdist takes ATen's `_euclidean_dist` matmul form
    sqrt(clamp([-2x, |x|^2, 1] . [e, 1, |e|^2]^T, 0)); argmin = first minimum."""
    x = x.astype(f32)
    x_norm = (x * x).sum(-1, keepdims=True, dtype=f32)
    e_norm = (E * E).sum(-1, keepdims=True, dtype=f32)
    x_ = np.concatenate([f32(-2) * x, x_norm, np.ones_like(x_norm)], -1)
    e_ = np.concatenate([E, np.ones_like(e_norm), e_norm], -1)
    d = np.sqrt(np.maximum((x_ @ e_.T).astype(f32), 0))
    return d.argmin(-1)


class MimiOracle:
    """MimiModel in streaming mode (compression.py:105-433) for a batch of B streams."""

    def __init__(self, state_dict, cfg, num_codebooks: int = 8):
        sd = {k: _np(v) for k, v in state_dict.items()}
        self.cfg = cfg
        self.n_q = num_codebooks
        c = cfg
        conv = lambda p, s=1, pad="constant", bias=True: StreamingConv1d(sd[p + ".weight"], sd[p + ".bias"] if bias else None, s, pad)
        # encoder (seanet.py:169-236)
        self.enc: List = [("conv", conv("encoder.model.0.conv.conv"))]
        idx = 1
        for ratio in reversed(c.ratios):
            self.enc.append(("res", conv(f"encoder.model.{idx}.block.1.conv.conv"), conv(f"encoder.model.{idx}.block.3.conv.conv")))
            idx += 2
            self.enc.append(("elu_conv", conv(f"encoder.model.{idx}.conv.conv", ratio)))
            idx += 1
        idx += 1
        self.enc.append(("elu_conv", conv(f"encoder.model.{idx}.conv.conv")))
        # decoder (seanet.py:315-388)
        self.dec: List = [("conv", conv("decoder.model.0.conv.conv"))]
        idx = 1
        for ratio in c.ratios:
            idx += 1
            p = f"decoder.model.{idx}.convtr.convtr"
            self.dec.append(("elu_convtr", StreamingConvTranspose1d(sd[p + ".weight"], sd[p + ".bias"], ratio)))
            idx += 1
            self.dec.append(("res", conv(f"decoder.model.{idx}.block.1.conv.conv"), conv(f"decoder.model.{idx}.block.3.conv.conv")))
            idx += 1
        idx += 1
        self.dec.append(("elu_conv", conv(f"decoder.model.{idx}.conv.conv")))
        mk = lambda name: [TransformerLayer(sd, f"{name}.transformer.layers.{l}", c.tr_num_heads, c.tr_context, c.tr_max_period)
                           for l in range(c.tr_num_layers)]
        self.enc_tr, self.dec_tr = mk("encoder_transformer"), mk("decoder_transformer")
        s = c.resample_stride
        self.down = StreamingConv1d(sd["downsample.conv.conv.conv.weight"], None, s, "replicate")     # resample.py:14-65
        self.up = StreamingConvTranspose1d(sd["upsample.convtr.convtr.convtr.weight"], None, s, depthwise=True)  # :68-119
        # quantiser (vq.py:76-84; core_vq.py:178-186)
        self.in_proj = [sd[f"quantizer.{p}.input_proj.weight"][:, :, 0].astype(f32) for p in ("rvq_first", "rvq_rest")]
        self.out_proj = [sd[f"quantizer.{p}.output_proj.weight"][:, :, 0].astype(f32) for p in ("rvq_first", "rvq_rest")]
        self.codebooks = []
        for k in range(c.q_n_q):
            p = (f"quantizer.rvq_first.vq.layers.{k}._codebook." if k < c.q_n_q_semantic
                 else f"quantizer.rvq_rest.vq.layers.{k - c.q_n_q_semantic}._codebook.")
            usage = np.maximum(sd[p + "cluster_usage"].astype(f32), f32(1e-5))
            self.codebooks.append((sd[p + "embedding_sum"].astype(f32) / usage[:, None]).astype(f32))
        self.exec_mask: Optional[np.ndarray] = None
        self.B = 0

    # ---- streaming lifecycle (streaming.py:110-211) ----------------------------------------------
    def _modules(self):
        for seq in (self.enc, self.dec):
            for item in seq:
                for m in item[1:]:
                    yield m
        yield self.down
        yield self.up
        for l in self.enc_tr + self.dec_tr:
            yield l

    def streaming(self, B: int):
        self.B = B
        self.exec_mask = np.ones(B, bool)
        for m in self._modules():
            m.init_state(B)

    def reset_streaming(self, mask: Optional[np.ndarray] = None):
        mask = np.ones(self.B, bool) if mask is None else np.asarray(mask, bool)
        self.exec_mask[mask] = True                      # streaming.py:43-44
        for m in self._modules():
            m.reset(mask)

    def set_exec_mask(self, mask):
        self.exec_mask = np.asarray(mask, bool).copy()

    # ---- SEANet --------------------------------------------------------------------------------
    def _run(self, seq, x):
        em = self.exec_mask
        for item in seq:
            kind = item[0]
            if kind == "conv":
                x = item[1](x, em)
            elif kind == "elu_conv" or kind == "elu_convtr":
                x = item[1](elu(x), em)
            elif kind == "res":                          # seanet.py:90-93
                v = item[2](elu(item[1](elu(x), em)), em)
                x = (x + v).astype(f32)
        return x

    def _transformer(self, layers, x):
        y = x.transpose(0, 2, 1)                         # conv_layout (transformer.py:971-983)
        for l in layers:
            y = l(y, self.exec_mask)
        return y.transpose(0, 2, 1).astype(f32)

    # ---- quantiser (vq.py:269-287; core_vq.py:507-528) -------------------------------------------
    def quantize(self, emb: np.ndarray) -> np.ndarray:
        B, C, T = emb.shape
        codes = np.zeros((B, self.n_q, T), np.int64)
        nsem = self.cfg.q_n_q_semantic
        for part, (k0, k1) in enumerate(((0, min(nsem, self.n_q)), (nsem, self.n_q))):
            if k1 <= k0:
                continue
            x = np.einsum("oc,bct->bto", self.in_proj[part], emb, optimize=True).astype(f32)   # [B, T, D]
            residual = x.reshape(B * T, -1)
            for k in range(k0, k1):
                idx = cdist_argmin(residual, self.codebooks[k])
                residual = (residual - self.codebooks[k][idx]).astype(f32)
                codes[:, k, :] = idx.reshape(B, T)
        return codes

    def decode_latent(self, codes: np.ndarray) -> np.ndarray:
        B, K, T = codes.shape
        nsem = self.cfg.q_n_q_semantic
        out = None
        for part, (k0, k1) in enumerate(((0, min(nsem, K)), (nsem, K))):
            if k1 <= k0:
                continue
            q = np.zeros((B, T, self.cfg.q_dimension), f32)
            for k in range(k0, k1):
                q = (q + self.codebooks[k][codes[:, k, :]]).astype(f32)
            y = np.einsum("od,btd->bot", self.out_proj[part], q, optimize=True).astype(f32)
            out = y if out is None else (out + y).astype(f32)
        return out

    # ---- public API (compression.py:338-433) -----------------------------------------------------
    def encode_to_latent(self, x: np.ndarray) -> np.ndarray:
        fs = self.cfg.frame_size
        assert x.shape[-1] % fs == 0 and x.shape[-1] > 0
        outs = []
        for f in range(x.shape[-1] // fs):                # frame by frame == the streaming contract
            e = self._run(self.enc, x[..., f * fs:(f + 1) * fs].astype(f32))
            e = self._transformer(self.enc_tr, e)
            outs.append(self.down(e, self.exec_mask))
        return np.concatenate(outs, -1)

    def encode(self, x: np.ndarray) -> np.ndarray:
        return self.quantize(self.encode_to_latent(x))

    def decode(self, codes: np.ndarray) -> np.ndarray:
        outs = []
        for f in range(codes.shape[-1]):
            e = self.decode_latent(codes[..., f:f + 1])
            e = self.up(e, self.exec_mask)
            e = self._transformer(self.dec_tr, e)
            outs.append(self._run(self.dec, e))
        return np.concatenate(outs, -1)
