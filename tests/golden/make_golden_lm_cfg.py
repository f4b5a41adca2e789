"""Golden vectors for `LMGen.step` with classifier-free guidance, sum-conditioning and extra heads,
produced by RUNNING THE REFERENCE (build container only: `python tests/golden/make_golden_lm_cfg.py`).

lm_cfg.npz - tiny Moshi LM (moshi_amd.config.tiny_lm_config), bf16, seeded weights re-drawn from the stored seed, B=2, greedy,
`support_out_of_sync=True`, an exec-mask schedule with one partial reset.  Scenarios (prefix):
  a_  cfg_coef 2.0, cfg_is_masked_until [1, 3]                       (lm.py:713-721)
  b_  cfg_coef 1.5, cfg_is_no_text                                   (lm.py:724-725, 731-732)
  c_  cfg_coef 3.0 + a `sum` condition [2B, 1, dim] through a ConditionFuser (lm.py:621-628, 399-400)
  d_  no CFG, a `sum` condition [B, 1, dim], and two extra heads read with `step_with_extra_heads` (lm.py:793-807)
lm_stt.npz - an ASR-style model (moshi_amd.config.tiny_stt_config: dep_q = 0, all 8 codebooks are input, text delayed by 2,
two extra heads), same schedule: tokens [B, 1, 1], text logits, extra-head probabilities.
For every step: the tokens returned, the logits every token was sampled from (recorded at `sample_token`, i.e. AFTER the
guidance combination), the sampled tokens, and for d_ the extra-head probabilities.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent


def run(lm_gen, codes, masks, reset_before, B, extra_heads=False):
    import moshi.models.lm as lm_mod
    out = {"tokens": [], "text_logits": [], "audio_logits": [], "text_tok": [], "audio_tok": [], "heads": []}
    rec = []
    orig = lm_mod.sample_token

    def sample_token(logits, *a, **k):
        tok = orig(logits, *a, **k)
        rec.append((logits.float().numpy().reshape(logits.shape[0], -1).copy(), tok.numpy().reshape(-1).copy()))
        return tok
    lm_mod.sample_token = sample_token
    try:
        with torch.no_grad(), lm_gen.streaming(B):
            for s in range(codes.shape[0]):
                rec.clear()
                if s in reset_before:
                    lm_gen.reset_streaming(torch.from_numpy(reset_before[s]))
                lm_gen.set_exec_mask(torch.from_numpy(masks[s]))
                if extra_heads:
                    o, heads = lm_gen.step_with_extra_heads(torch.from_numpy(codes[s]))
                    out["heads"].append(np.stack([h.float().numpy()[:, 0] for h in heads], 1))      # [B, n_heads, dim]
                else:
                    o = lm_gen.step(torch.from_numpy(codes[s]))
                out["tokens"].append(o.numpy().copy())
                out["text_logits"].append(rec[0][0]); out["text_tok"].append(rec[0][1])
                out["audio_logits"].append(np.stack([r[0] for r in rec[1:]], 1))
                out["audio_tok"].append(np.stack([r[1] for r in rec[1:]], 1))
    finally:
        lm_mod.sample_token = orig
    return {k: np.stack(v) for k, v in out.items() if v}


def main():
    from moshi.conditioners.base import ConditionFuser, ConditionType
    from moshi.models.lm import LMGen, LMModel
    from moshi_amd.config import tiny_lm_config
    from moshi_amd.weights import random_lm_state_dict
    cfg = tiny_lm_config()
    seed = 29
    sd = random_lm_state_dict(cfg, seed=seed)
    B, S = 2, 6
    g = torch.Generator().manual_seed(11)
    codes = torch.randint(0, cfg.card, (S, B, cfg.n_q - cfg.dep_q, 1), generator=g).numpy()
    masks = np.ones((S, B), bool)
    masks[2, 1] = False
    reset_before = {4: np.array([True, False])}
    cond2 = (0.5 * torch.randn(2 * B, 1, cfg.dim, generator=g)).to(torch.bfloat16)
    cond1 = (0.5 * torch.randn(B, 1, cfg.dim, generator=g)).to(torch.bfloat16)
    n_heads, hdim = 2, 6
    heads_w = [(torch.randn(hdim, cfg.dim, generator=g) / cfg.dim ** 0.5).to(torch.bfloat16) for _ in range(n_heads)]

    def model(fuser=None, extra=False):
        kw = cfg.reference_kwargs()
        if extra:
            kw.update(extra_heads_num_heads=n_heads, extra_heads_dim=hdim)
        lm = LMModel(**kw, fuser=fuser, device="cpu", dtype=torch.bfloat16)
        full = dict(sd)
        if extra:
            for i, w in enumerate(heads_w):
                full[f"extra_heads.{i}.weight"] = w
        lm.load_state_dict(full, strict=True)
        return lm.eval()

    def cond(t):
        return {"c": ConditionType(t, torch.ones(t.shape[:2], dtype=torch.bool))}

    out = {"seed": np.array([seed]), "codes": codes, "masks": masks, "reset_step": np.array([4]), "reset_mask": reset_before[4],
           "cond2": cond2.float().numpy(), "cond1": cond1.float().numpy(),
           "heads_w": np.stack([w.float().numpy() for w in heads_w]), "masked_until": np.array([1, 3])}
    common = dict(use_sampling=False, support_out_of_sync=True)
    runs = {
        "a": run(LMGen(model(), cfg_coef=2.0, cfg_is_masked_until=[1, 3], **common), codes, masks, reset_before, B),
        "b": run(LMGen(model(), cfg_coef=1.5, cfg_is_no_text=True, **common), codes, masks, reset_before, B),
        "c": run(LMGen(model(ConditionFuser({"sum": ["c"], "cross": []})), cfg_coef=3.0, condition_tensors=cond(cond2), **common),
                 codes, masks, reset_before, B),
        "d": run(LMGen(model(ConditionFuser({"sum": ["c"], "cross": []}), extra=True), condition_tensors=cond(cond1), **common),
                 codes, masks, reset_before, B, extra_heads=True),
    }
    for p, r in runs.items():
        out.update({f"{p}_{k}": v for k, v in r.items()})
    np.savez_compressed(HERE / "lm_cfg.npz", **out)
    print("lm_cfg.npz", {k: v.shape for k, v in out.items()})

    # ---- ASR-style model: no depformer
    from moshi_amd.config import tiny_stt_config
    scfg = tiny_stt_config()
    ssd = random_lm_state_dict(scfg, seed=37)
    lm = LMModel(**scfg.reference_kwargs(), device="cpu", dtype=torch.bfloat16)
    lm.load_state_dict(ssd, strict=True)
    lm.eval()
    assert lm.depformer is None
    codes8 = torch.randint(0, scfg.card, (S, B, scfg.n_q, 1), generator=g).numpy()
    import moshi.models.lm as lm_mod
    rec = []
    orig = lm_mod.sample_token

    def sample_token(logits, *a, **k):
        tok = orig(logits, *a, **k)
        rec.append((logits.float().numpy().reshape(logits.shape[0], -1).copy(), tok.numpy().reshape(-1).copy()))
        return tok
    lm_mod.sample_token = sample_token
    st = {"tokens": [], "text_logits": [], "text_tok": [], "heads": []}
    gen = LMGen(lm, **common)
    try:
        with torch.no_grad(), gen.streaming(B):
            for s_ in range(S):
                rec.clear()
                if s_ in reset_before:
                    gen.reset_streaming(torch.from_numpy(reset_before[s_]))
                gen.set_exec_mask(torch.from_numpy(masks[s_]))
                o, heads = gen.step_with_extra_heads(torch.from_numpy(codes8[s_]))
                st["tokens"].append(o.numpy().copy()); st["text_logits"].append(rec[0][0]); st["text_tok"].append(rec[0][1])
                st["heads"].append(np.stack([h.float().numpy()[:, 0] for h in heads], 1))
    finally:
        lm_mod.sample_token = orig
    sout = {"seed": np.array([37]), "codes": codes8, "masks": masks, "reset_step": np.array([4]), "reset_mask": reset_before[4]}
    sout.update({k: np.stack(v) for k, v in st.items()})
    np.savez_compressed(HERE / "lm_stt.npz", **sout)
    print("lm_stt.npz", {k: v.shape for k, v in sout.items()})


if __name__ == "__main__":
    sys.path.insert(0, "/root/reference/moshi")
    sys.path.insert(0, str(HERE.parent.parent))
    main()
