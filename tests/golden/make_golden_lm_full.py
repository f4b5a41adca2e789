"""Golden vectors for `LMGen.step` AT THE BENCHMARK DEPTH, produced by RUNNING THE REFERENCE (build container only).

lm_full.npz: Moshi-7B as `loaders._lm_kwargs` builds it (32 temporal layers, dim 4096, context 3000, depformer 6 x 1024 x 8
steps), bf16 on the CPU, weights re-drawn from the stored seed by moshi_amd.weights.random_lm_state_dict (the model bench.py
times), B = 2, greedy, 4 steps.  Step 0 executes row 0 only (exec mask), so the two rows sit at different stream offsets
for the remaining steps.  Stored: user codes, masks, ring outputs, the sampled tokens, and the text / audio logits as the bf16
bit patterns the reference produced.  Needs ~35 GB of host memory and a few minutes (15.4 GB of weights drawn + loaded).

    PYTHONPATH=/root/reference/moshi NO_TORCH_COMPILE=1 NO_CUDA_GRAPH=1 python tests/golden/make_golden_lm_full.py
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
SEED = 4242


def main():
    sys.path.insert(0, "/root/reference/moshi")
    sys.path.insert(0, str(HERE.parent.parent))
    sys.path.insert(0, str(HERE))
    from make_golden_lm import _run
    from moshi.models.lm import LMGen, LMModel
    from moshi_amd.config import LMConfig
    from moshi_amd.weights import random_lm_state_dict
    torch.set_num_threads(8)
    cfg = LMConfig()
    t0 = time.time()
    sd = random_lm_state_dict(cfg, seed=SEED)
    print("weights drawn", time.time() - t0, flush=True)
    with torch.device("meta"):
        lm = LMModel(**cfg.reference_kwargs(), dtype=torch.bfloat16)
    lm.load_state_dict(sd, strict=True, assign=True)
    lm.eval()
    print("model built", time.time() - t0, flush=True)
    B, S = 2, 4
    g = torch.Generator().manual_seed(11)
    codes = torch.randint(0, cfg.card, (S, B, cfg.n_q - cfg.dep_q, 1), generator=g).numpy()
    masks = np.ones((S, B), bool)
    masks[0, 1] = False
    greedy = _run(LMGen(lm, use_sampling=False, support_out_of_sync=True), lm, codes, masks, {}, B, False)
    print("steps run", time.time() - t0, flush=True)

    def bf16_bits(a):
        return (np.ascontiguousarray(a, np.float32).view(np.uint32) >> 16).astype(np.uint16)
    out = {"codes": codes, "masks": masks, "seed": np.array([SEED]), "g_tokens": greedy["tokens"], "g_text_tok": greedy["text_tok"],
           "g_audio_tok": greedy["audio_tok"], "g_text_logits_bf16": bf16_bits(greedy["text_logits"]),
           "g_audio_logits_bf16": bf16_bits(greedy["audio_logits"])}
    np.savez_compressed(HERE / "lm_full.npz", **out)
    print("lm_full.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
