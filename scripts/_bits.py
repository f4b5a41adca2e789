"""Greedy LM run (32 sessions, 2 temporal layers, the 7B widths) with taps; dumps tokens + logits for a bit-for-bit comparison of two builds."""
import sys, types, torch, numpy as np, hashlib
sys.path.insert(0, '.')
from moshi_amd.config import LMConfig
from moshi_amd.lm import LMGen, LMModel
from moshi_amd.weights import random_lm_state_dict
dev = torch.device("cuda", 0)
cfg = LMConfig(num_layers=2, context=64)
sd = random_lm_state_dict(cfg, seed=7, device=dev)
gen = LMGen(LMModel(sd, cfg, device=dev, max_batch=32), use_sampling=False, support_out_of_sync=True)
g = torch.Generator(device="cpu").manual_seed(3)
h = hashlib.sha1()
with gen.streaming(32):
    for s in range(6):
        codes = torch.randint(0, cfg.card, (32, 8, 1), generator=g).to(dev)
        out, tl, al = gen.step_with_taps(codes)
        for t in (out, tl, al):
            h.update(t.cpu().numpy().tobytes())
print("sha1 of tokens + text logits + audio logits over 6 greedy steps:", h.hexdigest())
