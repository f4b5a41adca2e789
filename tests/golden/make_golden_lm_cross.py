"""Golden vectors for `LMGen.step` on a model with cross-attention layers (second half), produced by RUNNING THE
REFERENCE (build container only: `python tests/golden/make_golden_lm_cross.py`).

lm_cross.npz - tiny Moshi LM with `cross_attention=True` (every temporal layer: self-attention, then
`x + cross_attention(norm_cross(x), src, src)`, then the gated FFN; transformer.py:727-732, 779-797), bf16, seeded weights
re-drawn from the stored seed, B=2, greedy, `support_out_of_sync=True`, the exec-mask schedule with one partial reset of
lm_cfg.npz.  Scenarios (prefix):
  e_  no CFG; `cross` condition [B, 5, dim] + a `sum` condition [B, 1, dim] through a ConditionFuser (base.py:392-421)
  f_  cfg_coef 2.0; `cross` condition [2B, 3, dim] built from TWO named tensors (concatenated along time) with
      `cross_attention_pos_emb=True`, scale 0.5 (base.py:402-408)
Recorded per step as in lm_cfg.npz (make_golden_lm_cfg.run)."""
from __future__ import annotations

import sys
from dataclasses import replace
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))


def main():
    from make_golden_lm_cfg import run
    from moshi.conditioners.base import ConditionFuser, ConditionType
    from moshi.models.lm import LMGen, LMModel
    from moshi_amd.config import tiny_lm_config
    from moshi_amd.weights import random_lm_state_dict
    cfg = replace(tiny_lm_config(), cross_attention=True)
    seed = 53
    sd = random_lm_state_dict(cfg, seed=seed)
    B, S = 2, 6
    g = torch.Generator().manual_seed(17)
    codes = torch.randint(0, cfg.card, (S, B, cfg.n_q - cfg.dep_q, 1), generator=g).numpy()
    masks = np.ones((S, B), bool)
    masks[2, 1] = False
    reset_before = {4: np.array([True, False])}
    cross_e = (0.7 * torch.randn(B, 5, cfg.dim, generator=g)).to(torch.bfloat16)
    sum_e = (0.5 * torch.randn(B, 1, cfg.dim, generator=g)).to(torch.bfloat16)
    cross_f1 = (0.7 * torch.randn(2 * B, 2, cfg.dim, generator=g)).to(torch.bfloat16)
    cross_f2 = (0.7 * torch.randn(2 * B, 1, cfg.dim, generator=g)).to(torch.bfloat16)

    def model(fuser):
        lm = LMModel(**cfg.reference_kwargs(), fuser=fuser, device="cpu", dtype=torch.bfloat16)
        lm.load_state_dict(dict(sd), strict=True)
        return lm.eval()

    def ct(t):
        return ConditionType(t, torch.ones(t.shape[:2], dtype=torch.bool))
    common = dict(use_sampling=False, support_out_of_sync=True)
    out = {"seed": np.array([seed]), "codes": codes, "masks": masks, "reset_step": np.array([4]), "reset_mask": reset_before[4],
           "cross_e": cross_e.float().numpy(), "sum_e": sum_e.float().numpy(), "cross_f1": cross_f1.float().numpy(),
           "cross_f2": cross_f2.float().numpy()}
    runs = {
        "e": run(LMGen(model(ConditionFuser({"sum": ["s"], "cross": ["x"]})), condition_tensors={"s": ct(sum_e), "x": ct(cross_e)},
                       **common), codes, masks, reset_before, B),
        "f": run(LMGen(model(ConditionFuser({"sum": [], "cross": ["x1", "x2"]}, cross_attention_pos_emb=True,
                                            cross_attention_pos_emb_scale=0.5)),
                       cfg_coef=2.0, condition_tensors={"x1": ct(cross_f1), "x2": ct(cross_f2)}, **common),
                 codes, masks, reset_before, B),
    }
    for p, r in runs.items():
        out.update({f"{p}_{k}": v for k, v in r.items()})
    np.savez_compressed(HERE / "lm_cross.npz", **out)
    print("lm_cross.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
