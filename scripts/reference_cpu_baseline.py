"""The REFERENCE's own PyTorch CPU path (kyutai-labs/moshi, imported read-only from /root/reference) timed on THIS host: the
recipe of scripts/moshi_benchmark.py:76-100 (zeros / noise chunk -> mimi.encode -> lm_gen.step -> mimi.decode per 80 ms
frame), BASELINE.md section 3.  The reference cannot travel to the GPU box (no copy of its sources is kept in this repository),
so bench.py's `cpu_baseline` times the numpy port there and quotes this file's output beside it (profiles/r03_logs/
reference_cpu_baseline.json; labelled with the host it was measured on).

    PYTHONPATH=/root/reference/moshi NO_TORCH_COMPILE=1 NO_CUDA_GRAPH=1 python scripts/reference_cpu_baseline.py
"""
import json
import os
import platform
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent


def p50(xs):
    return float(sorted(xs)[len(xs) // 2])


def main():
    sys.path.insert(0, "/root/reference/moshi")
    sys.path.insert(0, str(ROOT))
    from moshi.models import loaders
    from moshi.models.lm import LMGen, LMModel
    from moshi_amd.config import LMConfig
    from moshi_amd.weights import random_lm_state_dict
    threads = os.cpu_count()
    torch.set_num_threads(threads)
    import re
    m = re.search(r"model name\s*:\s*(.*)", open("/proc/cpuinfo").read())
    out = {"host": {"cpu": m.group(1).strip() if m else platform.processor(), "cores": threads, "torch": torch.__version__,
                    "where": "the build container (not the GPU box)"},
           "recipe": "scripts/moshi_benchmark.py:76-100 on device='cpu', NO_TORCH_COMPILE=1, random-init weights, 0.1*N(0,1) PCM"}
    torch.manual_seed(1234)
    mimi = loaders.get_mimi(None, None, "cpu", num_codebooks=8)
    for m in mimi.modules():
        if hasattr(m, "embedding_sum"):
            m.embedding_sum.normal_()
    for B in (1, 8):
        enc, dec = [], []
        with torch.no_grad(), mimi.streaming(B):
            for i in range(5 + 30):
                x = 0.1 * torch.randn(B, 1, 1920)
                t0 = time.perf_counter(); codes = mimi.encode(x); t1 = time.perf_counter(); mimi.decode(codes); t2 = time.perf_counter()
                if i >= 5:
                    enc.append(t1 - t0); dec.append(t2 - t1)
        out[f"mimi_b{B}"] = {"encode_p50_ms": 1e3 * p50(enc), "decode_p50_ms": 1e3 * p50(dec)}
        print(f"mimi B={B}", out[f"mimi_b{B}"], flush=True)
    cfg = LMConfig()
    sd = random_lm_state_dict(cfg, seed=4242)
    with torch.device("meta"):
        lm = LMModel(**cfg.reference_kwargs(), dtype=torch.bfloat16)
    lm.load_state_dict(sd, strict=True, assign=True)
    lm.eval()
    gen = LMGen(lm, use_sampling=True, temp=0.8, temp_text=0.7, top_k=250, top_k_text=25)
    ts = []
    with torch.no_grad(), gen.streaming(1):
        for i in range(1 + 5):
            codes = torch.randint(0, cfg.card, (1, 8, 1))
            t0 = time.perf_counter(); gen.step(codes); ts.append(time.perf_counter() - t0)
    out["lm_7b_bf16_b1"] = {"step_p50_ms": 1e3 * p50(ts[1:]), "steps_timed": len(ts) - 1, "first_step_ms": 1e3 * ts[0]}
    print("lm", out["lm_7b_bf16_b1"], flush=True)
    frame_s = (out["mimi_b1"]["encode_p50_ms"] + out["mimi_b1"]["decode_p50_ms"] + out["lm_7b_bf16_b1"]["step_p50_ms"]) / 1e3
    out["duplex_b1_frames_per_s"] = 1.0 / frame_s
    dst = ROOT / "profiles" / "r03_logs" / "reference_cpu_baseline.json"
    dst.write_text(json.dumps(out, indent=1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
