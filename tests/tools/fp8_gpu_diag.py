"""GPU diagnostic: error of the fp8 engine against the fp8 oracle on the tiny model, per sampling site (max / mean relative to
max|ref|), for several seeds and batch tilings - to separate rounding-boundary noise from a systematic difference."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from moshi_amd.config import tiny_lm_config  # noqa: E402
from moshi_amd.lm import LMGen, LMModel  # noqa: E402
from moshi_amd.weights import quantize_lm_state_dict, quantize_lm_state_dict_fp8, random_lm_state_dict  # noqa: E402
from oracle.lm_oracle import LMOracle  # noqa: E402

cfg = tiny_lm_config()
import sys as _sys
MODES = _sys.argv[1].split(",") if len(_sys.argv) > 1 else ["bf16", "fp8"]
for mode in MODES:
    for B in (2, 18):
        mx, mean, n_tok, n_bad = [], [], 0, 0
        for seed in (92, 108):
            sd = random_lm_state_dict(cfg, seed=seed)
            if mode == "fp8":
                sd = quantize_lm_state_dict_fp8(sd)
            elif mode == "int8":
                sd = quantize_lm_state_dict(sd)
            gen = LMGen(LMModel(sd, cfg, device="cuda", max_batch=B), use_sampling=False, support_out_of_sync=True)
            orc = LMOracle(sd, cfg)
            orc.streaming(B)
            rng = np.random.default_rng(seed)
            with gen.streaming(B):
                for s in range(3):
                    codes = rng.integers(0, cfg.card, (B, 8, 1))
                    oo, (otl, oal, ott, oat) = orc.step(codes, use_sampling=False, support_out_of_sync=True)
                    forced = np.concatenate([ott[:, None], oat], 1)
                    out, tl, al = gen.step_with_taps(torch.from_numpy(codes).cuda(), forced_tokens=torch.from_numpy(forced).cuda())
                    tl, al = tl.cpu().numpy(), al.cpu().numpy()
                    for b in range(B):
                        for a, r in [(tl[b], otl[b])] + [(al[b, k], oal[b, k]) for k in range(cfg.dep_q)]:
                            sc = np.abs(r).max() + 1e-6
                            d = np.abs(a - r)
                            mx.append(d.max() / sc); mean.append(d.mean() / sc)
                            n_tok += 1
                            n_bad += int(a.argmax() != r.argmax())
        mx, mean = np.array(mx), np.array(mean)
        print(f"{mode:5s} B={B:2d}: max-rel  p50 {np.median(mx):.4f} p99 {np.quantile(mx, .99):.4f} worst {mx.max():.4f} | "
              f"mean-rel p50 {np.median(mean):.4f} worst {mean.max():.4f} | argmax differs {n_bad}/{n_tok}", flush=True)

if "wide" in MODES or len(_sys.argv) > 2:
    from moshi_amd.config import LMConfig
    wcfg = LMConfig(num_layers=2, context=64)
    for mode in ("bf16", "fp8"):
        sd = random_lm_state_dict(wcfg, seed=10)
        if mode == "fp8":
            sd = quantize_lm_state_dict_fp8(sd)
        B = 3
        gen = LMGen(LMModel(sd, wcfg, device="cuda", max_batch=B), use_sampling=False, support_out_of_sync=True)
        orc = LMOracle(sd, wcfg)
        orc.streaming(B)
        rng = np.random.default_rng(10)
        mx, mean, scales = [], [], []
        with gen.streaming(B):
            for s in range(2):
                codes = rng.integers(0, wcfg.card, (B, 8, 1))
                oo, (otl, oal, ott, oat) = orc.step(codes, use_sampling=False, support_out_of_sync=True)
                forced = np.concatenate([ott[:, None], oat], 1)
                out, tl, al = gen.step_with_taps(torch.from_numpy(codes).cuda(), forced_tokens=torch.from_numpy(forced).cuda())
                tl, al = tl.cpu().numpy(), al.cpu().numpy()
                for b in range(B):
                    for a, r in [(tl[b], otl[b])] + [(al[b, k], oal[b, k]) for k in range(wcfg.dep_q)]:
                        sc = np.abs(r).max() + 1e-6
                        d = np.abs(a - r)
                        mx.append(d.max() / sc); mean.append(d.mean() / sc); scales.append(sc)
        mx, mean = np.array(mx), np.array(mean)
        print(f"wide {mode:5s} B=3: max-rel p50 {np.median(mx):.4f} worst {mx.max():.4f} | mean-rel p50 {np.median(mean):.4f} "
              f"worst {mean.max():.4f} | max|ref| range {min(scales):.2f}..{max(scales):.2f}", flush=True)
        del gen, orc, sd
