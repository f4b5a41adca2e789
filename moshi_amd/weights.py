"""State-dict layout of the two models (names/shapes exactly as the reference checkpoints carry them,
SURVEY.md Appendix A; reference: seanet.py:169-236,315-388, transformer.py:407-418, vq.py:76-84,
core_vq.py:158-160, lm.py:135-232) and seeded synthetic weights of that layout.

There is no network and no released checkpoint in this environment, so benchmarks and parity tests use
synthetic weights: drawn on the CPU generator from a seed (bit-reproducible across machines for one torch
build), with magnitudes that keep activations O(1).  The golden-vector generator loads the very same tensors
into the reference implementation (`load_state_dict`), which is what makes its outputs comparable.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch

from .config import LMConfig, MimiConfig

Spec = List[Tuple[str, Tuple[int, ...], str]]  # (name, shape, kind)


def mimi_state_spec(cfg: MimiConfig) -> Spec:
    spec: Spec = []

    def conv(prefix: str, cout: int, cin: int, k: int, bias: bool = True):
        spec.append((prefix + ".weight", (cout, cin, k), f"fan:{cin * k}"))
        if bias:
            spec.append((prefix + ".bias", (cout,), f"fan:{cin * k}"))

    nf, nr = cfg.n_filters, len(cfg.ratios)
    # encoder
    conv("encoder.model.0.conv.conv", nf, cfg.channels, cfg.kernel_size)
    idx, mult = 1, 1
    for ratio in reversed(cfg.ratios):
        ch, hid = mult * nf, mult * nf // cfg.compress
        conv(f"encoder.model.{idx}.block.1.conv.conv", hid, ch, cfg.residual_kernel_size)
        conv(f"encoder.model.{idx}.block.3.conv.conv", ch, hid, 1)
        idx += 2
        conv(f"encoder.model.{idx}.conv.conv", 2 * ch, ch, 2 * ratio)
        idx += 1
        mult *= 2
    idx += 1
    conv(f"encoder.model.{idx}.conv.conv", cfg.dimension, mult * nf, cfg.last_kernel_size)
    # decoder
    mult = 2 ** nr
    conv("decoder.model.0.conv.conv", mult * nf, cfg.dimension, cfg.kernel_size)
    idx = 1
    for ratio in cfg.ratios:
        cin, cout = mult * nf, mult * nf // 2
        idx += 1
        spec.append((f"decoder.model.{idx}.convtr.convtr.weight", (cin, cout, 2 * ratio), f"fan:{cin * 2}"))
        spec.append((f"decoder.model.{idx}.convtr.convtr.bias", (cout,), f"fan:{cin * 2}"))
        idx += 1
        conv(f"decoder.model.{idx}.block.1.conv.conv", cout // cfg.compress, cout, cfg.residual_kernel_size)
        conv(f"decoder.model.{idx}.block.3.conv.conv", cout, cout // cfg.compress, 1)
        idx += 1
        mult //= 2
    idx += 1
    conv(f"decoder.model.{idx}.conv.conv", cfg.channels, nf, cfg.last_kernel_size)
    # transformers
    d, ff = cfg.tr_d_model, cfg.tr_dim_feedforward
    for name in ("encoder_transformer", "decoder_transformer"):
        for l in range(cfg.tr_num_layers):
            p = f"{name}.transformer.layers.{l}"
            spec.append((p + ".self_attn.in_projs.0.weight", (3 * d, d), f"fan:{d}"))
            spec.append((p + ".self_attn.out_projs.0.weight", (d, d), f"fan:{d}"))
            spec.append((p + ".norm1.weight", (d,), "norm_w"))
            spec.append((p + ".norm1.bias", (d,), "norm_b"))
            spec.append((p + ".norm2.weight", (d,), "norm_w"))
            spec.append((p + ".norm2.bias", (d,), "norm_b"))
            spec.append((p + ".linear1.weight", (ff, d), f"fan:{d}"))
            spec.append((p + ".linear2.weight", (d, ff), f"fan:{ff}"))
            spec.append((p + ".layer_scale_1.scale", (d,), "layer_scale"))
            spec.append((p + ".layer_scale_2.scale", (d,), "layer_scale"))
    # resampling
    s = cfg.resample_stride
    spec.append(("downsample.conv.conv.conv.weight", (cfg.dimension, cfg.dimension, 2 * s), f"fan:{cfg.dimension * 2 * s}"))
    spec.append(("upsample.convtr.convtr.convtr.weight", (cfg.dimension, 1, 2 * s), "fan:2"))
    # quantiser
    D, bins = cfg.q_dimension, cfg.q_bins
    for part, n in (("rvq_first", cfg.q_n_q_semantic), ("rvq_rest", cfg.q_n_q - cfg.q_n_q_semantic)):
        spec.append((f"quantizer.{part}.input_proj.weight", (D, cfg.dimension, 1), f"fan:{cfg.dimension}"))
        spec.append((f"quantizer.{part}.output_proj.weight", (cfg.dimension, D, 1), f"fan:{D}"))
        for k in range(n):
            p = f"quantizer.{part}.vq.layers.{k}._codebook."
            spec.append((p + "_initialized", (1,), "one"))
            spec.append((p + "cluster_usage", (bins,), "usage"))
            spec.append((p + "embedding_sum", (bins, D), f"codebook:{k}"))
    return spec


def lm_state_spec(cfg: LMConfig) -> Spec:
    spec: Spec = []
    d, dd = cfg.dim, cfg.depformer_dim
    h, dh = cfg.ffn_hidden, cfg.depformer_ffn_hidden
    for i in range(cfg.n_q):
        spec.append((f"emb.{i}.weight", (cfg.card + 1, d), f"emb:{d}"))
    spec.append(("text_emb.weight", (cfg.text_card + 1, d), f"emb:{d}"))
    spec.append(("text_linear.weight", (cfg.text_card, d), f"fan:{d}"))
    spec.append(("out_norm.alpha", (1, 1, d), "alpha"))
    for l in range(cfg.num_layers):
        p = f"transformer.layers.{l}"
        spec.append((p + ".self_attn.in_projs.0.weight", (3 * d, d), f"fan:{d}"))
        spec.append((p + ".self_attn.out_projs.0.weight", (d, d), f"fan:{d}"))
        spec.append((p + ".norm1.alpha", (1, 1, d), "alpha"))
        spec.append((p + ".norm2.alpha", (1, 1, d), "alpha"))
        spec.append((p + ".gating.linear_in.weight", (2 * h, d), f"fan:{d}"))
        spec.append((p + ".gating.linear_out.weight", (d, h), f"fan:{h}"))
        if cfg.cross_attention:                # transformer.py:727-731: a second MHA + its own nn.LayerNorm (weight and bias)
            spec.append((p + ".cross_attention.in_projs.0.weight", (3 * d, d), f"fan:{d}"))
            spec.append((p + ".cross_attention.out_projs.0.weight", (d, d), f"fan:{d}"))
            spec.append((p + ".norm_cross.weight", (d,), "norm_w"))
            spec.append((p + ".norm_cross.bias", (d,), "norm_b"))
    for k in range(cfg.dep_q):
        spec.append((f"depformer_in.{k}.weight", (dd, d), f"fan:{d}"))
    for k in range(cfg.dep_q - 1):
        spec.append((f"depformer_emb.{k}.weight", (cfg.card + 1, dd), f"emb:{dd}"))
    if cfg.dep_q > 0:                      # "No-Depformer --- e.g., an ASR model" (lm.py:187-221): none of the depth weights exist
        spec.append(("depformer_text_emb.weight", (cfg.text_card + 1, dd), f"emb:{dd}"))
    for l in range(cfg.depformer_num_layers if cfg.dep_q > 0 else 0):
        p = f"depformer.layers.{l}"
        for k in range(cfg.dep_q):
            spec.append((p + f".self_attn.in_projs.{k}.weight", (3 * dd, dd), f"fan:{dd}"))
            spec.append((p + f".self_attn.out_projs.{k}.weight", (dd, dd), f"fan:{dd}"))
        spec.append((p + ".norm1.alpha", (1, 1, dd), "alpha"))
        spec.append((p + ".norm2.alpha", (1, 1, dd), "alpha"))
        for k in range(cfg.dep_q):
            spec.append((p + f".gating.{k}.linear_in.weight", (2 * dh, dd), f"fan:{dd}"))
            spec.append((p + f".gating.{k}.linear_out.weight", (dd, dh), f"fan:{dh}"))
    for k in range(cfg.dep_q):
        spec.append((f"linears.{k}.weight", (cfg.card, dd), f"fan:{dd}"))
    for i in range(cfg.extra_heads_num_heads):          # lm.py:224-226
        spec.append((f"extra_heads.{i}.weight", (cfg.extra_heads_dim, d), f"fan:{d}"))
    return spec


def _draw(shape, kind: str, gen: torch.Generator, device, dtype) -> torch.Tensor:
    """One synthetic tensor.  Drawn on `device` when it is a GPU (fast for the 7B benchmark model)."""
    def rnd(fn, *a):
        return fn(*a, generator=gen, device=device, dtype=torch.float32)
    if kind.startswith("fan:"):
        fan = int(kind[4:])
        t = (rnd(torch.rand, shape) * 2 - 1) * math.sqrt(3.0 / fan)   # unit-gain uniform
    elif kind.startswith("emb:"):
        t = rnd(torch.randn, shape) / math.sqrt(int(kind[4:]))
    elif kind.startswith("codebook"):
        t = rnd(torch.randn, shape)
    elif kind == "usage":
        t = rnd(torch.rand, shape) * 1.5 + 0.5
    elif kind == "norm_w" or kind == "alpha":
        t = 1.0 + 0.1 * rnd(torch.randn, shape)
    elif kind == "norm_b":
        t = 0.05 * rnd(torch.randn, shape)
    elif kind == "layer_scale":
        t = 0.3 + 0.05 * rnd(torch.randn, shape)   # larger than the 0.01 init so the transformer matters in tests
    elif kind == "one":
        t = torch.ones(shape, device=device, dtype=torch.float32)
    else:
        raise ValueError(kind)
    return t.to(dtype)


def random_state_dict(spec: Spec, seed: int, device="cpu", dtype=torch.float32) -> Dict[str, torch.Tensor]:
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    return {name: _draw(shape, kind, gen, device, dtype) for name, shape, kind in spec}


def random_mimi_state_dict(cfg: MimiConfig, seed: int = 1234, device="cpu") -> Dict[str, torch.Tensor]:
    return random_state_dict(mimi_state_spec(cfg), seed, device, torch.float32)


def random_lm_state_dict(cfg: LMConfig, seed: int = 4242, device="cpu", dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    return random_state_dict(lm_state_spec(cfg), seed, device, dtype)


# --------------------------------------------------------------------------------------------------------------------
# checkpoint key normalisation (what the reference does in its state-dict load hooks)
# --------------------------------------------------------------------------------------------------------------------
def normalize_lm_state_dict(sd: Dict[str, torch.Tensor], cfg: LMConfig) -> Dict[str, torch.Tensor]:
    """Released Moshi checkpoints store multi-step attention projections fused over the steps
    (`...self_attn.in_proj_weight` [mult*3*dim, dim], `...self_attn.out_proj.weight` [mult*dim, dim]); the reference splits
    them into `in_projs.{i}.weight` / `out_projs.{i}.weight` in `StreamingMultiheadAttention._load_hook`
    (modules/transformer.py:422-446; `scripts/import_rust.py:91-101` shows the depformer layout).  Same mapping here, so a
    checkpoint's state dict can be handed to `LMModel` as is.  mult = 1 for the temporal transformer, dep_q for the depformer."""
    out: Dict[str, torch.Tensor] = {}
    sources = {"in_proj_weight": "in_projs.{i}.weight", "in_proj.weight": "in_projs.{i}.weight",
               "in_proj.lora_A.weight": "in_projs.{i}.lora_A.weight", "in_proj.lora_B.weight": "in_projs.{i}.lora_B.weight",
               "out_proj.weight": "out_projs.{i}.weight",
               "out_proj.lora_A.weight": "out_projs.{i}.lora_A.weight", "out_proj.lora_B.weight": "out_projs.{i}.lora_B.weight"}
    for key, val in sd.items():
        hit = None
        for suffix in ("", "_scb"):                       # `_scb`: the row scales of a quantised checkpoint travel with their weight
            for src, dst in sources.items():
                for attn in (".self_attn.", ".cross_attention."):
                    if key.endswith(attn + src + suffix):
                        hit = (key[: -len(src + suffix)], dst + suffix)
                        break
                if hit:
                    break
            if hit:
                break
        if hit is None:
            out[key] = val
            continue
        prefix, dst = hit
        mult = cfg.dep_q if prefix.startswith("depformer.") else 1
        assert val.shape[0] % mult == 0, f"{key}: leading dimension {val.shape[0]} is not a multiple of {mult} steps"
        parts = val.view(mult, -1, *val.shape[1:])
        for i in range(mult):
            out[prefix + dst.format(i=i)] = parts[i]
    return out


def normalize_mimi_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Legacy codebook buffer names (`inited`, `cluster_size`, `embed_avg` / `embed_sum`) -> current ones, as
    `EuclideanCodebook._load_from_state_dict` does (quantization/core_vq.py:162-176); the transformer projections get the
    same un-fusing as the LM (mult = 1)."""
    ren = {"inited": "_initialized", "cluster_size": "cluster_usage", "embed_avg": "embedding_sum", "embed_sum": "embedding_sum"}
    out: Dict[str, torch.Tensor] = {}
    for key, val in sd.items():
        head, _, leaf = key.rpartition(".")
        if head.endswith("._codebook") and leaf in ren:
            key = head + "." + ren[leaf]
        for src, dst in (("in_proj_weight", "in_projs.0.weight"), ("in_proj.weight", "in_projs.0.weight"), ("out_proj.weight", "out_projs.0.weight")):
            if key.endswith(".self_attn." + src):
                key = key[: -len(src)] + dst
                break
        out[key] = val
    return out


# --------------------------------------------------------------------------------------------------------------------
# int8 weights (the reference's `quantize=True`: utils/quantize.py QLinear, bitsandbytes int8_vectorwise_quant)
# --------------------------------------------------------------------------------------------------------------------
def is_lm_linear_weight(key: str) -> bool:
    """The nn.Linear weights `replace_linear_with_qlinear` converts (lm.py:242-243, transformer.py:885-888): every linear of
    the temporal and depth transformers, `depformer_in`, `linears`, `text_linear`.  Embeddings and norms stay bf16."""
    if not key.endswith(".weight"):
        return False
    return (".self_attn.in_projs." in key or ".self_attn.out_projs." in key or ".linear_in.weight" in key
            or ".linear_out.weight" in key or key.startswith("depformer_in.") or key.startswith("linears.")
            or key == "text_linear.weight")


E4M3_MAX = 448.0


def quantize_lm_state_dict_fp8(sd: Dict[str, torch.Tensor], input_scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """fp8 linears for the fp8 MFMA path (BASELINE.json configs[4]): per-output-row scaling, `weight` =
    e4m3fn(W / weight_scale) with weight_scale = absmax_row / 448 (fp32, `<key>_scale`), and a static scalar activation
    scale `<linear>.input_scale` (a calibration constant; powers of two cost no precision).  W ~= weight * weight_scale."""
    out: Dict[str, torch.Tensor] = {}
    for key, w in sd.items():
        if not is_lm_linear_weight(key) or w.dtype == torch.float8_e4m3fn:
            out[key] = w
            continue
        wf = w.detach().float()
        absmax = wf.abs().amax(dim=1)
        scale = torch.where(absmax > 0, absmax / E4M3_MAX, torch.ones_like(absmax))
        out[key] = (wf / scale[:, None]).clamp_(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn)
        out[key + "_scale"] = scale.to(torch.float32)
        if input_scale != 1.0:
            out[key[: -len(".weight")] + ".input_scale"] = torch.tensor([input_scale], dtype=torch.float32)
    return out


def quantize_lm_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Row-wise absmax int8, as `QLinear.__init__` does it: the weight goes to fp16, `CB = round(W * 127 / absmax_row)`
    (int8) is stored under the weight's key and `SCB = absmax_row` (fp32) under `<key>_scb` (utils/quantize.py:17-22;
    the key suffix is the one the reference's load hook handles, transformer.py:435-446).  W ~= CB * SCB / 127."""
    out: Dict[str, torch.Tensor] = {}
    for key, w in sd.items():
        if not is_lm_linear_weight(key) or w.dtype == torch.int8:
            out[key] = w
            continue
        w16 = w.detach().to(torch.float16).float()
        absmax = w16.abs().amax(dim=1)
        scale = torch.where(absmax > 0, 127.0 / absmax, torch.zeros_like(absmax))
        out[key] = torch.round(w16 * scale[:, None]).clamp_(-127, 127).to(torch.int8)
        out[key + "_scb"] = absmax.to(torch.float32)
    return out


# --------------------------------------------------------------------------------------------------------------------
# LoRA adapters (modules/lora.py): merged into the base weights at load
# --------------------------------------------------------------------------------------------------------------------
def fuse_lora_state_dict(sd: Dict[str, torch.Tensor], lora_sd: Dict[str, torch.Tensor], scaling: float) -> Dict[str, torch.Tensor]:
    """`replace_lora_with_linear` (modules/lora.py:26-43): W' = W + scaling * (B @ A) for every linear that has
    `<linear>.lora_A.weight` [rank, in] and `<linear>.lora_B.weight` [out, rank] in `lora_sd`, evaluated like the reference in
    the weights' own dtype (bf16 matmul, scale, add).  The engine always runs the merged weights (the reference can also keep
    the adapter unfused, W x + scaling * B (A x): the same function up to rounding).  Unknown adapter keys are an error."""
    out = dict(sd)
    used = set()
    for key in list(lora_sd):
        if not key.endswith(".lora_A.weight"):
            continue
        stem = key[: -len(".lora_A.weight")]
        kb, kw = stem + ".lora_B.weight", stem + ".weight"
        if kb not in lora_sd:
            raise RuntimeError(f"LoRA weights carry {key} without {kb}")
        if kw not in out:
            raise RuntimeError(f"unexpected_keys in the lora weights: {[key, kb]}")     # loaders.py:508-511
        w = out[kw]
        a, b = lora_sd[key].to(w.dtype), lora_sd[kb].to(w.dtype)
        if a.shape[1] != w.shape[1] or b.shape[0] != w.shape[0] or a.shape[0] != b.shape[1]:
            raise RuntimeError(f"LoRA shapes {tuple(b.shape)} x {tuple(a.shape)} do not fit {kw} {tuple(w.shape)}")
        out[kw] = w + scaling * (b @ a)
        used.update((key, kb))
    extra = [k for k in lora_sd if k not in used and not k.endswith(".frozen_W.weight")]
    if extra:
        raise RuntimeError(f"unexpected_keys in the lora weights: {extra}")
    return out
