"""LM half of bench.py: Moshi-7B construction, staggered session starts, dominant-kernel roofline, CPU proxy."""
from __future__ import annotations

import ctypes as C
import json
import os
import time
from pathlib import Path

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0


def replicated_state_dict(draw, spec, dtype, dev):
    """One process per GPU: rank 0 draws the (7.7 B parameter, bf16, seeded) weights on its GPU and replicates them to the
    other ranks with RCCL broadcasts in 1 GiB buckets (moshi_amd/dist.py) - the deployment's only collective, at load, as in
    SURVEY.md 8e.  Single process, or MMI_BENCH_NO_BCAST=1: every rank draws the same seeded weights itself."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1 or os.environ.get("MMI_BENCH_NO_BCAST"):
        return draw()
    from moshi_amd.dist import broadcast_state_dict
    sd = draw() if dist.get_rank() == 0 else None
    return broadcast_state_dict(sd, spec, dtype, dev, src=0)


def make_lm(dev, B, args, streaming=True):
    from moshi_amd.config import LMConfig
    from moshi_amd.lm import LMGen, LMModel
    from moshi_amd.weights import lm_state_spec, random_lm_state_dict
    cfg = LMConfig(kv_cache_dtype=getattr(args, "kv", "bf16"))
    if args.lm_layers:
        cfg.num_layers = args.lm_layers
    sd = replicated_state_dict(lambda: random_lm_state_dict(cfg, seed=4242, device=dev), lm_state_spec(cfg), torch.bfloat16, dev)
    quant = getattr(args, "quant", "none")
    if quant in ("q8", "fp8"):                                  # quantise tensor by tensor (frees the bf16 copy as it goes)
        from moshi_amd.weights import is_lm_linear_weight, quantize_lm_state_dict, quantize_lm_state_dict_fp8
        fn = quantize_lm_state_dict if quant == "q8" else quantize_lm_state_dict_fp8
        for k in [k for k in sd if is_lm_linear_weight(k)]:
            sd.update(fn({k: sd.pop(k)}))
    lm = LMModel(sd, cfg, device=dev, max_batch=B)
    del sd
    torch.cuda.empty_cache()
    if not streaming:
        return lm
    gen = LMGen(lm, use_sampling=True, temp=0.8, temp_text=0.7, top_k=250, top_k_text=25, seed=1234 + int(os.environ.get("RANK", "0")))
    gen.streaming_forever(B)
    return gen


def stagger(mimi, lm_gen, step_fn, B, frames_apart, dev):
    """sessions start staggered (row b has run frames_apart*b frames when timing starts)."""
    if frames_apart <= 0 or B == 1:
        return 0
    n = frames_apart * (B - 1)
    rows = torch.arange(B, device=dev)
    for f in range(n):
        mask = rows >= (B - 1 - f // frames_apart)
        if mimi is not None:
            mimi.set_exec_mask(mask)
        lm_gen.set_exec_mask(mask)
        step_fn()
    ones = torch.ones(B, dtype=torch.bool, device=dev)
    if mimi is not None:
        mimi.set_exec_mask(ones)
    lm_gen.set_exec_mask(ones)
    return n


def lm_step_algorithmic_bytes(cfg, L_per_row):   # bf16 weights
    """weights once per step + per-stream KV read/write (bf16)."""
    d, dd, h, dh = cfg.dim, cfg.depformer_dim, cfg.ffn_hidden, cfg.depformer_ffn_hidden
    per_layer = 3 * d * d + d * d + 2 * h * d + d * h
    dep = cfg.depformer_num_layers * cfg.dep_q * (3 * dd * dd + dd * dd + 2 * dh * dd + dd * dh)
    nw = cfg.num_layers * per_layer + cfg.text_card * d + dep + cfg.dep_q * dd * d + cfg.dep_q * cfg.card * dd
    kv = sum(2 * cfg.num_layers * 2 * d * (min(int(L), cfg.context) + 1) for L in L_per_row)
    return 2 * nw + kv


def roofline_lm(lm_gen, step_fn, args, sync):
    """Dominant kernel = the temporal FFN linear_in GEMM (184.5 MB of weights per launch, 32 launches per step).
    Timed live with hipEvents on the launch stream over `steps` un-graphed steps (include/moshi_mi.h profile tap)."""
    lib, h = lm_gen._lib, lm_gen.lm_model._handle
    sync()
    lib.check(lib.mmi_lm_profile_begin(h))
    for _ in range(max(4, min(args.steps, 20))):
        step_fn()
    mean_ms, n, nbytes, name = C.c_double(), C.c_int64(), C.c_int64(), C.c_char_p()
    lib.check(lib.mmi_lm_profile_end(h, C.byref(mean_ms), C.byref(n), C.byref(nbytes), C.byref(name)))
    ach = nbytes.value / (mean_ms.value * 1e-3) / 1e9 if mean_ms.value > 0 else 0.0
    kname = name.value.decode() if name.value else ""
    out = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
           "traffic": None, "kernel": kname, "avg_launch_ms": mean_ms.value,
           "launches_timed": n.value, "algorithmic_bytes_per_launch": nbytes.value}
    # secondary figure (north_star: "achieved MFMA/HBM fraction"): the same launch's matrix-core rate.  The GEMM is
    # 2 * N * K * B flops with N = 2 * ffn_hidden rows, K = dim, B sessions; dense peaks from MI355X_MICROARCH.md.
    cfg = lm_gen.lm_model.config
    flops = 2.0 * (2 * cfg.ffn_hidden) * cfg.dim * lm_gen._batch
    quant = getattr(args, "quant", "none")
    mfma_peak = 2500.0                       # TFLOP/s: bf16 MFMA; the non-scaled fp8 MFMA (K=16/32) runs at the bf16 rate
    tf = flops / (mean_ms.value * 1e-3) / 1e12 if mean_ms.value > 0 else 0.0
    out["mfma"] = {"achieved": tf, "peak": mfma_peak, "unit": "TFLOP/s", "frac": tf / mfma_peak,
                   "instruction": "v_mfma_f32_32x32x16_fp8_fp8" if quant == "fp8" else "v_mfma_f32_32x32x16_bf16",
                   "note": "decode GEMM at B sessions: arithmetic intensity ~B flop/byte, far below the ~310 flop/byte ridge"}
    # HBM bytes per launch from the PMC counters: rocprofv3 --pmc cannot run inside the benchmark (and crashes on this
    # process), so the figure is the committed measurement of the same kernel, shape and batch.
    pmc = Path(__file__).resolve().parent / "profiles" / "pmc_dominant_kernel.json"
    if pmc.exists() and lm_gen._batch == 32:
        rec = json.loads(pmc.read_text())
        if rec["kernel"] in kname and rec["algorithmic_bytes_per_launch"] == nbytes.value:
            out["traffic"] = rec["traffic_bytes_per_launch"]
            out["traffic_source"] = "profiles/pmc_dominant_kernel.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
    return out


def cpu_baseline_duplex(mimi_base, args):
    """`port` baseline for the full frame on the host cores: the numpy oracles.  Mimi is timed directly (mimi_base);
    the LM oracle is timed at reduced depth (1 and 2 temporal layers, full depformer, B=1) and extrapolated linearly to
    32 layers - a layer-scaled proxy, as BASELINE.md section 3 allows when the full 7B fp32 oracle (30 GB) is too heavy."""
    from moshi_amd.config import LMConfig
    from moshi_amd.weights import random_lm_state_dict
    from oracle.lm_oracle import LMOracle
    times = {}
    for nl in (1, 2):
        cfg = LMConfig(num_layers=nl, context=64)
        o = LMOracle(random_lm_state_dict(cfg, seed=1, device="cpu"), cfg)
        o.streaming(1)
        codes = np.zeros((1, 8, 1), np.int64)
        o.step(codes, use_sampling=False)
        t0 = time.perf_counter()
        n = 2
        for _ in range(n):
            o.step(codes, use_sampling=False)
        times[nl] = (time.perf_counter() - t0) / n
        del o
    full_layers = 32 if not args.lm_layers else args.lm_layers
    per_layer = max(times[2] - times[1], 0.0)
    lm_s = times[1] + (full_layers - 1) * per_layer
    mimi_s = 1.0 / mimi_base["value"]
    return {"value": 1.0 / (lm_s + mimi_s), "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
            "sample": (f"B=1: Mimi oracle {mimi_s*1e3:.0f} ms/frame ({mimi_base['sample']}); LM oracle timed at 1 and 2 "
                       f"temporal layers + full depformer ({times[1]:.2f} s, {times[2]:.2f} s per step) and extrapolated "
                       f"linearly to {full_layers} layers = {lm_s:.2f} s/step (layer-scaled proxy)")}
