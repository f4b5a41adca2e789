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


def stagger(mimi, lm_gen, step_fn, B, frames_apart, dev, before_mask=None):
    """sessions start staggered (row b has run frames_apart*b frames when timing starts).
    before_mask: called before every mask change (the pipelined step joins its streams there)."""
    if frames_apart <= 0 or B == 1:
        return 0
    n = frames_apart * (B - 1)
    rows = torch.arange(B, device=dev)

    def set_mask(mask):
        if before_mask is not None:
            before_mask()
        if mimi is not None:
            mimi.set_exec_mask(mask)
        lm_gen.set_exec_mask(mask)
    for f in range(n):
        if f % frames_apart == 0:
            set_mask(rows >= (B - 1 - f // frames_apart))
        step_fn()
    set_mask(torch.ones(B, dtype=torch.bool, device=dev))
    return n


def lm_step_algorithmic_bytes(cfg, L_per_row, quant="none", kv="bf16"):
    """weights once per step (2 B bf16; 1 B + a 4-byte scale per output row for int8 / fp8) + per-stream KV
    read of the valid positions and write of the new one (2 B bf16, 1 B fp8)."""
    d, dd, h, dh = cfg.dim, cfg.depformer_dim, cfg.ffn_hidden, cfg.depformer_ffn_hidden
    per_layer = 3 * d * d + d * d + 2 * h * d + d * h
    rows_layer = 3 * d + d + 2 * h + d
    dep = cfg.depformer_num_layers * cfg.dep_q * (3 * dd * dd + dd * dd + 2 * dh * dd + dd * dh)
    dep_rows = cfg.depformer_num_layers * cfg.dep_q * (3 * dd + dd + 2 * dh + dd)
    nw = cfg.num_layers * per_layer + cfg.text_card * d + dep + cfg.dep_q * dd * d + cfg.dep_q * cfg.card * dd
    nrows = cfg.num_layers * rows_layer + cfg.text_card + dep_rows + cfg.dep_q * dd + cfg.dep_q * cfg.card
    s_kv = 1 if kv == "fp8" else 2
    kvb = sum(s_kv * cfg.num_layers * 2 * d * (min(int(L), cfg.context) + 1) for L in L_per_row)
    wbytes = 2 * nw if quant == "none" else nw + 4 * nrows
    return wbytes + kvb


def roofline_lm(lm_gen, step_fn, args, sync, kv_rows=None):
    """Dominant kernel = the temporal FFN linear_in GEMM (184.5 MB of weights per launch, 32 launches per step).
    Timed live with hipEvents on the launch stream over `steps` un-graphed steps (include/moshi_mi.h profile tap)."""
    lib, h = lm_gen._lib, lm_gen.lm_model._handle
    sync()
    lib.check(lib.mmi_lm_profile_begin(h))
    for _ in range(max(4, min(args.steps, 20))):
        step_fn()
    sync()
    from moshi_amd import _capi
    n_prof = max(4, min(args.steps, 20))
    site_txt = _capi.read_text(lambda buf, cap: lib.mmi_lm_profile_sites(h, buf, cap))
    mean_ms, n, nbytes, name = C.c_double(), C.c_int64(), C.c_int64(), C.c_char_p()
    lib.check(lib.mmi_lm_profile_end(h, C.byref(mean_ms), C.byref(n), C.byref(nbytes), C.byref(name)))
    ach = nbytes.value / (mean_ms.value * 1e-3) / 1e9 if mean_ms.value > 0 else 0.0
    kname = name.value.decode() if name.value else ""
    out = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
           "traffic": None, "kernel": kname, "avg_launch_ms": mean_ms.value,
           "launches_timed": n.value, "algorithmic_bytes_per_launch": nbytes.value}
    # every site of the LM step, live: hipEvent pairs around each op of the same un-graphed steps (dispatch gaps included, so a
    # few us above the rocprofv3 kernel durations of profiles/*_sites.csv).  GB/s = packed weight bytes of the GEMM / time;
    # the attention's bytes are the valid K and V rows of the sessions at their depth in the middle of these steps.
    cfg_ = lm_gen.lm_model.config
    sites = {}
    for line in site_txt.splitlines():
        f = line.split("\t")
        if len(f) < 4 or int(f[1]) == 0:
            continue
        site, ops, tot_ms, wbytes = f[0], int(f[1]), float(f[2]), int(f[3])
        us = 1e3 * tot_ms / ops
        rec = {"ops_per_step": ops / n_prof, "us_per_op": us, "us_per_step": 1e3 * tot_ms / n_prof}
        if site == "L.attn" and kv_rows is not None:
            s_kv = 1 if getattr(args, "kv", "bf16") == "fp8" else 2
            wbytes = int(sum(2 * cfg_.dim * s_kv * (min(L, cfg_.context) + 1) for L in kv_rows))
        if wbytes > 0:
            rec.update({"bytes_per_op": wbytes, "GBps": wbytes / us / 1e3, "frac": wbytes / us / 1e3 / HBM_PEAK_GBS})
        sites[site] = rec
    out["sites"] = sites
    # the largest site by time per step (at the mid-run depth the decode attention outweighs the dominant GEMM)
    timed = {k: dict(v) for k, v in sites.items() if "frac" in v and v.get("ops_per_step", 0) > 0}
    if "L.ffn_in" in timed and mean_ms.value > 0:
        # this site's ops carry a SECOND event pair (the dominant-kernel tap above): rank it by that tap's own figure instead
        t = timed["L.ffn_in"]
        t.update({"us_per_op": 1e3 * mean_ms.value, "us_per_step": 1e3 * mean_ms.value * t["ops_per_step"],
                  "GBps": t["bytes_per_op"] / (1e3 * mean_ms.value) / 1e3, "frac": t["bytes_per_op"] / (1e3 * mean_ms.value) / 1e3 / HBM_PEAK_GBS})
    if timed:
        big = max(timed, key=lambda k: timed[k]["us_per_step"])
        out["largest_site"] = {"site": big, "kernel": {"L.attn": "k_lm_attn_wave"}.get(big, "see mmi_lm_launch_list"), **timed[big],
                               "note": "live hipEvents around each op of un-graphed steps (a few us of dispatch per op included)"}
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
    # HBM bytes per launch from the PMC counters: a counter pass cannot run inside the timed benchmark (it serialises every
    # dispatch), so the figure is the committed measurement of the same kernel, shape and batch - since round 3 collected on
    # THIS library inside the LM step (scripts/gpu_pmc.sh: eager launches under rocprofv3 --pmc), before that on a standalone launcher.
    pmc = Path(__file__).resolve().parent / "profiles" / "pmc_dominant_kernel.json"
    if pmc.exists() and lm_gen._batch == 32:
        doc = json.loads(pmc.read_text())
        for rec in doc.get("kernels", [doc]):
            if rec["kernel"] in kname and rec["algorithmic_bytes_per_launch"] == nbytes.value:
                out["traffic"] = rec["traffic_bytes_per_launch"]
                where = "inside the LM step of this library (scripts/gpu_pmc.sh)" if rec.get("measured_in_step") else "standalone launcher"
                out["traffic_source"] = ("profiles/pmc_dominant_kernel.json: a COMMITTED measurement of this kernel, shape and batch (rocprofv3 --pmc "
                                         f"FETCH_SIZE x2 + WRITE_SIZE, separate passes, {where}; " + ", ".join(rec.get("sources", [])) + ") - not collected in this run")
                out["traffic_measured_in_this_run"] = False
                break
        att = doc.get("attention")
        if att and "largest_site" in out and out["largest_site"]["site"] == "L.attn":
            out["largest_site"]["traffic"] = {"fetch_bytes_per_launch": att[0]["fetch_bytes_corrected"], "what": att[0]["what"],
                                              "note": att[0]["note"], "sources": att[0]["sources"], "measured_in_this_run": False}
    return out


def _time_lm_steps(gen, codes, n_warm, n, dev):
    """ms of each of n LMGen.step calls after n_warm untimed ones (device events around every step, one synchronisation)."""
    for _ in range(n_warm):
        gen.step(codes)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for a, b in evs:
        a.record(); gen.step(codes); b.record()
    torch.cuda.synchronize(dev)
    wall = 1e3 * (time.perf_counter() - t0) / n
    return wall, sorted(a.elapsed_time(b) for a, b in evs)


def extra_full_context(lm_gen, user_codes, B, dev, steps=10):
    """`full_context` of the default line: LMGen.step ALONE on the benchmark's handle with every session's ring full (SURVEY.md
    8d: "B=32, L=3000: 14.75 + 50.3 GB -> 8.1 ms" floor) - the sessions are moved to position `context` (mmi_lm_seek), so every
    step reads all 3000 slots of every ring.  Runs after everything else (it leaves the sessions there)."""
    cfg = lm_gen.lm_model.config
    lm_gen.seek([cfg.context] * B)
    wall, lat = _time_lm_steps(lm_gen, user_codes, 3, steps, dev)
    nbytes = lm_step_algorithmic_bytes(cfg, [cfg.context] * B)
    ach = nbytes / (wall * 1e-3) / 1e9
    return {"workload": f"LMGen.step alone, {B} sessions, every KV ring full ({cfg.context} positions, bf16)", "steps": steps,
            "ms_per_step": wall, "p50_ms_per_step": lat[len(lat) // 2], "algorithmic_bytes": nbytes, "achieved_GBps": ach,
            "frac": ach / HBM_PEAK_GBS}


def extra_c3(dev, args, steps=30):
    """`c3` of the default line (BASELINE configs[2]: Moshi-7B bf16 LMGen.step, batch 1 - the single real-time session): a
    second handle built for ONE session (16-row MFMA tile), moved to the midpoint of SURVEY 8d's 300-step run, sampled."""
    import copy
    a1 = copy.copy(args)
    gen = make_lm(dev, 1, a1)
    cfg = gen.lm_model.config
    gen.seek([150])
    codes = torch.randint(0, cfg.card, (1, cfg.n_q - cfg.dep_q, 1), device=dev)
    wall, lat = _time_lm_steps(gen, codes, 5, steps, dev)
    nbytes = lm_step_algorithmic_bytes(cfg, [150 + 5 + steps // 2])
    ach = nbytes / (wall * 1e-3) / 1e9
    out = {"workload": "Moshi-7B bf16 LMGen.step, ONE session (BASELINE configs[2]), ring 150 + deep (midpoint of a 300-step run)",
           "steps": steps, "ms_per_step": wall, "p50_ms_per_step": lat[len(lat) // 2], "p95_ms_per_step": lat[min(len(lat) - 1, int(0.95 * len(lat)))],
           "budget_ms": 80.0, "algorithmic_bytes": nbytes, "achieved_GBps": ach, "frac": ach / HBM_PEAK_GBS}
    try:        # the per-site table of the one-session step, live (hipEvents around every op of un-graphed steps)
        a1.steps = 8
        r = roofline_lm(gen, lambda: gen.step(codes), a1, lambda: torch.cuda.synchronize(dev), kv_rows=[150 + 5 + steps + 4])
        out["sites"] = {k: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in r["sites"].items()}
    except Exception as e:      # noqa: BLE001 - a diagnostic, never worth the line
        out["sites_error"] = repr(e)
    del gen
    torch.cuda.empty_cache()
    return out


def extra_c5(args):
    """`c5` of the default line (BASELINE configs[4]: q8 weights on the 8-bit matrix core, 64 sessions per GPU): the same duplex
    step - Mimi encode -> LMGen.step -> Mimi decode, pipelined, mid-run ring depth - with row-wise int8 linears x row-wise int8
    activations (the reference's bitsandbytes arithmetic) and the e4m3 KV ring, measured by a child process running this very
    benchmark with `--batch 64 --quant q8 --kv fp8` (its own handles, its own memory; the parent's stay resident), plus the same
    with the reference's bf16 ring.  The child's whole-step roofline fraction travels with it."""
    import subprocess
    import sys
    out = {}
    for key, kv in (("fp8_ring", "fp8"), ("bf16_ring", "bf16")):
        cmd = [sys.executable, str(Path(__file__).resolve().parent / "bench.py"), "--no-cpu-baseline", "--no-extras", "--batch", "64",
               "--quant", "q8", "--kv", kv, "--steps", str(min(args.steps, 40)), "--warmup", str(min(args.warmup, 8))]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l][-1]
            d = json.loads(line)
            step = (d.get("roofline") or {}).get("step") or {}
            out[key] = {"ms_per_step": d["ms_per_step"], "p50_ms_per_step": d.get("p50_ms_per_step"), "frames_per_s": d["value"],
                        "algorithmic_bytes": step.get("algorithmic_bytes"), "achieved_GBps": step.get("achieved"), "frac": step.get("frac")}
        except Exception as e:      # noqa: BLE001 - an extra, never worth the line
            out[key] = {"error": repr(e)[:300]}
    out["workload"] = ("duplex step (Mimi encode -> LMGen.step -> Mimi decode, pipelined), 64 sessions, Moshi-7B with row-wise int8 linears x int8 "
                       "activations on v_mfma_i32_*_i8 (BASELINE configs[4]), mid-run ring depth; fp8_ring = e4m3 KV ring, bf16_ring = the reference's")
    return out


def extra_c2(args):
    """`c2` of the default line (BASELINE configs[1]: Mimi streaming encode + RVQ + decode, 8 streams, one GPU): a child process running
    this very benchmark with `--workload mimi --batch 8` (fp32 codec, the reference's own width - what keeps the codes bit-exact)."""
    import subprocess
    import sys
    cmd = [sys.executable, str(Path(__file__).resolve().parent / "bench.py"), "--no-cpu-baseline", "--no-extras", "--workload", "mimi", "--batch", "8",
           "--steps", str(max(args.steps, 60)), "--warmup", str(max(args.warmup, 12))]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l][-1]
        d = json.loads(line)
        return {"workload": "Mimi streaming encode + RVQ + decode, 8 streams, fp32 (BASELINE configs[1])", "ms_per_step": d["ms_per_step"],
                "p50_ms_per_step": d.get("p50_ms_per_step"), "frames_per_s": d["value"], "steps": d.get("steps")}
    except Exception as e:      # noqa: BLE001 - an extra, never worth the line
        return {"error": repr(e)[:300]}


def _mem_available_gib():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / (1 << 20)
    except OSError:
        pass
    return 0.0


def cpu_lm_measured(dev, timed=3):
    """The LM oracle at the model's FULL depth (32 temporal layers, full depformer and text head, fp32 numpy on the host cores,
    B = 1): one warm-up step, then the median of `timed` steps.  The weights are the benchmark's own (same seed), drawn on the
    GPU and copied to the host (15 GB bf16 -> 30 GB fp32).  Needs a host with >= 64 GB free."""
    from moshi_amd.config import LMConfig
    from moshi_amd.weights import random_lm_state_dict
    from oracle.lm_oracle import LMOracle
    cfg = LMConfig()
    t0 = time.perf_counter()
    sd = random_lm_state_dict(cfg, seed=4242, device=dev)
    sd = {k: v.cpu() for k, v in sd.items()}
    torch.cuda.empty_cache()
    o = LMOracle(sd, cfg)
    del sd
    o.streaming(1)
    t_build = time.perf_counter() - t0
    rng = np.random.default_rng(0)
    o.step(rng.integers(0, cfg.card, (1, 8, 1)), use_sampling=False)
    ts = []
    for _ in range(timed):
        codes = rng.integers(0, cfg.card, (1, 8, 1))
        t0 = time.perf_counter()
        o.step(codes, use_sampling=False)
        ts.append(time.perf_counter() - t0)
    del o
    return float(np.median(ts)), ts, t_build


def _host_info():
    ram = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemTotal"):
                ram = int(line.split()[1]) // (1 << 20)
    except OSError:
        pass
    return os.cpu_count(), ram


def _reference_quote():
    """The reference ITSELF cannot travel to the GPU box (no copy of its sources is kept here): its own CPU path, timed on the
    build container by scripts/reference_cpu_baseline.py (recipe of scripts/moshi_benchmark.py:76-100), is quoted beside the port."""
    rp = Path(__file__).resolve().parent / "profiles" / "r03_logs" / "reference_cpu_baseline.json"
    if not rp.exists():
        return None
    rd = json.loads(rp.read_text())
    return {"value": rd["duplex_b1_frames_per_s"], "unit": "frames/s", "cores": rd["host"]["cores"], "cpu": rd["host"]["cpu"],
            "measured_on": rd["host"].get("where", "build container"), "mimi_b1_ms": rd["mimi_b1"], "mimi_b8_ms": rd["mimi_b8"],
            "lm_7b_bf16_b1_step_ms": rd["lm_7b_bf16_b1"]["step_p50_ms"],
            "note": "kyutai-labs/moshi PyTorch CPU path, B=1, NOT measured in this run (committed: profiles/r03_logs/reference_cpu_baseline.json)"}


def cpu_baseline_duplex(mimi_base, args, layers=(1, 5), timed=5, dev=None):
    """`port` baseline for the full frame on the host cores: the numpy oracles (the reference itself cannot travel to the GPU
    box).  Mimi is timed directly (mimi_base); the LM oracle (fp32, B=1, full depformer and text head) is timed at two reduced
    depths - median of `timed` steps after one warm-up step each, the temporal stack (`forward_text`) also on its own clock -
    and extrapolated linearly to the model's 32 temporal layers with the per-layer cost of that stack: the layer-scaled proxy BASELINE.md section 3 allows when the full 7B fp32 oracle (30 GB) is too heavy for a
    default run.  The slope must be positive (a 32-layer model that costs what one layer costs is not a baseline)."""
    from moshi_amd.config import LMConfig
    from moshi_amd.weights import random_lm_state_dict
    from oracle.lm_oracle import LMOracle
    # the full 32-layer oracle, MEASURED, whenever the host can hold it; the layer-scaled proxy below is the
    # fallback SURVEY.md 8d describes for a host that cannot (or a run with --lm-layers / MMI_BENCH_CPU_PROXY=1)
    if dev is not None and not args.lm_layers and not os.environ.get("MMI_BENCH_CPU_PROXY") and _mem_available_gib() >= 64:
        lm_s, ts, t_build = cpu_lm_measured(dev)
        mimi_s = 1.0 / mimi_base["value"]
        cores, ram = _host_info()
        return {"value": 1.0 / (lm_s + mimi_s), "unit": "frames/s", "cores": cores, "host_ram_gib": ram, "kind": "port",
                "reference_on_build_host": _reference_quote(),
                "sample": (f"B=1: Mimi oracle {mimi_s*1e3:.0f} ms/frame ({mimi_base['sample']}); LM oracle (fp32 numpy): 32 layers, measured - "
                           f"the whole Moshi-7B step (32 temporal layers, text head, 8 x 6 depth-transformer layers), 1 warm-up + median of "
                           f"{len(ts)} steps = {lm_s:.3f} s ({', '.join(f'{t:.3f}' for t in ts)}); weights drawn on the GPU and widened to fp32 on "
                           f"the host in {t_build:.0f} s (not timed)")}
    times, text_times = {}, {}
    for nl in layers:
        cfg = LMConfig(num_layers=nl, context=64)
        o = LMOracle(random_lm_state_dict(cfg, seed=1, device="cpu"), cfg)
        o.streaming(1)
        ft = o.forward_text
        ft_t = []

        def timed_forward_text(tokens, _ft=ft, _acc=ft_t):      # the depth-dependent part, timed on its own: the 208 small
            t0 = time.perf_counter()                            # depformer GEMVs of a step are noisy under a threaded BLAS
            r = _ft(tokens)
            _acc.append(time.perf_counter() - t0)
            return r
        o.forward_text = timed_forward_text
        rng = np.random.default_rng(nl)
        o.step(rng.integers(0, cfg.card, (1, 8, 1)), use_sampling=False)
        ft_t.clear()
        ts = []
        for _ in range(timed):
            codes = rng.integers(0, cfg.card, (1, 8, 1))
            t0 = time.perf_counter()
            o.step(codes, use_sampling=False)
            ts.append(time.perf_counter() - t0)
        times[nl], text_times[nl] = float(np.median(ts)), float(np.median(ft_t))
        del o
    lo, hi = layers
    per_layer = (text_times[hi] - text_times[lo]) / (hi - lo)
    if not per_layer > 0:
        raise RuntimeError(f"cpu_baseline: non-positive per-layer cost ({text_times}); the host is too noisy for a baseline")
    full_layers = 32 if not args.lm_layers else args.lm_layers
    lm_s = times[lo] + (full_layers - lo) * per_layer
    mimi_s = 1.0 / mimi_base["value"]
    cores, ram = _host_info()
    ref = _reference_quote()
    return {"value": 1.0 / (lm_s + mimi_s), "unit": "frames/s", "cores": cores, "host_ram_gib": ram, "kind": "port", "reference_on_build_host": ref,
            "sample": (f"B=1: Mimi oracle {mimi_s*1e3:.0f} ms/frame ({mimi_base['sample']}); LM oracle (fp32 numpy) median of "
                       f"{timed} steps at {lo} and {hi} temporal layers + full depformer and text head ({times[lo]:.3f} s, "
                       f"{times[hi]:.3f} s per step; temporal stack alone {text_times[lo]*1e3:.0f} / {text_times[hi]*1e3:.0f} ms = "
                       f"{per_layer*1e3:.1f} ms per temporal layer) extrapolated linearly to "
                       f"{full_layers} layers = {lm_s:.2f} s/step (layer-scaled proxy)")}
