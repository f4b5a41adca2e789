"""Golden vectors for `LMGen.step`, produced by RUNNING THE REFERENCE (build container only; see make_golden.py).

lm_tiny.npz: tiny Moshi LM (moshi_amd.config.tiny_lm_config), bf16, synthetic seeded weights (re-drawn from the stored seed), B=3:
  * a greedy run of 7 steps with an exec-mask schedule and a partial reset (recipe of
    scripts/test_missing_data_lm.py: `LMGen(..., use_sampling=False, support_out_of_sync=True)`), and
  * a sampled run of 4 steps (temp .8/.7, top-k 20/10) whose Exp(1) draws are recorded at the reference's
    `multinomial` (sampling.py:40-47) so that the same noise can be replayed,
each with the user codes, the step outputs, the text / audio logits the tokens were sampled from, and the tokens.

lm_wide.npz: Moshi-7B's real layer shapes (dim 4096, 32 heads x 128, FFN 11264, text head 32000; depformer 1024 x 6 layers x
8 steps) with ONE temporal layer and context 16, bf16, seeded weights, B=2, a greedy run of 3 steps: pins RoPE at head
dim 128, the 32-head ring attention, the gated FFN width, the full-size depformer and both full-size heads.  Logits are
stored as the bf16 bit patterns the reference produced.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent


def _run(lm_gen, lm, codes, masks, reset_before, B, record_noise):
    from moshi.utils import sampling
    out = {"tokens": [], "text_logits": [], "audio_logits": [], "text_tok": [], "audio_tok": [], "noise": []}
    cur = {}
    lm_gen.on_text_logits_hook = lambda t: cur.__setitem__("tl", t.float().numpy()[:, 0, 0].copy())
    lm_gen.on_text_hook = lambda t: cur.__setitem__("tt", t.numpy().copy())
    lm_gen.on_audio_hook = lambda t: cur.__setitem__("at", t.numpy().copy())
    orig_fd = lm.forward_depformer

    def fd(k, seq, tout):
        lg = orig_fd(k, seq, tout)
        cur.setdefault("al", []).append(lg.float().numpy()[:, 0, 0].copy())
        return lg
    lm.forward_depformer = fd
    orig_mn = sampling.multinomial

    def mn(inp, num_samples, replacement=False, *, generator=None):
        # same draw as sampling.py:40-47, recorded
        inp_ = inp.reshape(-1, inp.shape[-1])
        q = torch.empty_like(inp_).exponential_(1, generator=generator)
        cur.setdefault("q", []).append(q.numpy().copy())
        return (inp_ / q).argmax(dim=-1, keepdim=True).reshape(*list(inp.shape[:-1]), -1)
    if record_noise:
        sampling.multinomial = mn
    try:
        with torch.no_grad(), lm_gen.streaming(B):
            for s in range(codes.shape[0]):
                cur.clear()
                if s in reset_before:
                    lm_gen.reset_streaming(torch.from_numpy(reset_before[s]))
                lm_gen.set_exec_mask(torch.from_numpy(masks[s]))
                o = lm_gen.step(torch.from_numpy(codes[s]))
                out["tokens"].append(o.numpy().copy())
                out["text_logits"].append(cur["tl"]); out["audio_logits"].append(np.stack(cur["al"], 1))
                out["text_tok"].append(cur["tt"]); out["audio_tok"].append(cur["at"])
                if record_noise:
                    kmax = max(q.shape[-1] for q in cur["q"])
                    nz = np.ones((B, len(cur["q"]), kmax), np.float32)
                    for i, q in enumerate(cur["q"]):
                        nz[:, i, :q.shape[-1]] = q
                    out["noise"].append(nz)
    finally:
        sampling.multinomial = orig_mn
        lm.forward_depformer = orig_fd
    return {k: np.stack(v) for k, v in out.items() if v}


def gen_lm_tiny():
    from moshi.models.lm import LMGen, LMModel
    from moshi_amd.config import tiny_lm_config
    from moshi_amd.weights import random_lm_state_dict
    cfg = tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=17)
    lm = LMModel(**cfg.reference_kwargs(), device="cpu", dtype=torch.bfloat16)
    missing = lm.load_state_dict(sd, strict=True)
    lm.eval()
    B = 3
    g = torch.Generator().manual_seed(3)
    S = 7
    codes = torch.randint(0, cfg.card, (S, B, cfg.n_q - cfg.dep_q, 1), generator=g).numpy()
    masks = np.ones((S, B), bool)
    masks[2, 1] = masks[3, 1] = False
    masks[4, 2] = False
    reset_before = {5: np.array([True, False, False])}
    greedy = _run(LMGen(lm, use_sampling=False, support_out_of_sync=True), lm, codes, masks, reset_before, B, False)
    S2 = 4
    codes2 = torch.randint(0, cfg.card, (S2, B, cfg.n_q - cfg.dep_q, 1), generator=g).numpy()
    torch.manual_seed(99)
    sampled = _run(LMGen(lm, use_sampling=True, temp=0.8, temp_text=0.7, top_k=20, top_k_text=10, support_out_of_sync=True),
                   lm, codes2, np.ones((S2, B), bool), {}, B, True)
    # the reference's None-during-delay behaviour without support_out_of_sync (lm.py:774-776)
    plain = LMGen(lm, use_sampling=False)
    nones = []
    with torch.no_grad(), plain.streaming(B):
        for s in range(3):
            nones.append(plain.step(torch.from_numpy(codes[s])) is None)
    out = {"codes": codes, "masks": masks, "reset_step": np.array([5]), "reset_mask": reset_before[5],
           "codes2": codes2, "none_pattern": np.array(nones)}
    out.update({"g_" + k: v for k, v in greedy.items()})
    out.update({"s_" + k: v for k, v in sampled.items()})
    out["seed"] = np.array([17])   # weights are re-drawn from the seed by moshi_amd.weights (not stored)
    np.savez_compressed(HERE / "lm_tiny.npz", **out)
    print("lm_tiny.npz", {k: v.shape for k, v in out.items() if not k.startswith("sd/")})


def gen_lm_wide():
    from moshi.models.lm import LMGen, LMModel
    from moshi_amd.config import LMConfig
    from moshi_amd.weights import random_lm_state_dict
    cfg = LMConfig(num_layers=1, context=16)
    sd = random_lm_state_dict(cfg, seed=23)
    lm = LMModel(**cfg.reference_kwargs(), device="cpu", dtype=torch.bfloat16)
    lm.load_state_dict(sd, strict=True)
    lm.eval()
    B, S = 2, 3
    g = torch.Generator().manual_seed(5)
    codes = torch.randint(0, cfg.card, (S, B, cfg.n_q - cfg.dep_q, 1), generator=g).numpy()
    greedy = _run(LMGen(lm, use_sampling=False, support_out_of_sync=True), lm, codes, np.ones((S, B), bool), {}, B, False)

    def bf16_bits(a):   # the logits are bf16 values widened to fp32: keep the upper 16 bits
        return (np.ascontiguousarray(a, np.float32).view(np.uint32) >> 16).astype(np.uint16)
    out = {"codes": codes, "seed": np.array([23]), "g_tokens": greedy["tokens"], "g_text_tok": greedy["text_tok"],
           "g_audio_tok": greedy["audio_tok"], "g_text_logits_bf16": bf16_bits(greedy["text_logits"]),
           "g_audio_logits_bf16": bf16_bits(greedy["audio_logits"])}
    np.savez_compressed(HERE / "lm_wide.npz", **out)
    print("lm_wide.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    sys.path.insert(0, "/root/reference/moshi")
    sys.path.insert(0, str(HERE.parent.parent))
    if "--wide-only" not in sys.argv:
        gen_lm_tiny()
    gen_lm_wide()
