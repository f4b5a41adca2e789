"""Benchmark of the 12.5 Hz full-duplex frame step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload duplex|mimi|lm] [--batch B]

One "step" = one pass of the hot path over one batch of synthetic input: B concurrent sessions each advance by
one 80 ms frame (Mimi encode -> LMGen.step -> Mimi decode).  `value` = frames/s summed over all ranks; sessions
shard data-parallel (one process per GPU, no data-path collective), so scaling is weak.
Prints ONE JSON line on rank 0 (contract in the task statement) incl. the `roofline` and `cpu_baseline` objects.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def trace(msg):
    if os.environ.get("MMI_BENCH_TRACE"):
        sys.stderr.write(f"[bench {time.perf_counter():.3f}] {msg}\n")
        sys.stderr.flush()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--workload", default="auto", choices=["auto", "duplex", "mimi", "lm", "served", "launchcheck"],
                    help="served: the duplex step driven through the session batcher with HOST PCM in / PCM + tokens out "
                         "(pinned staging, H2D + D2H and one synchronisation per step included): the PCIe-inclusive rate")
    ap.add_argument("--batch", type=int, default=32, help="sessions per GPU (BASELINE.json configs[3]: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stagger", type=int, default=8, help="frames between session starts (SURVEY.md 8d C4)")
    ap.add_argument("--quant", default="none", choices=["none", "q8", "fp8"],
                    help="BASELINE configs[4] weight formats: q8 = row-wise int8 linears x row-wise int8 activations on the int8 MFMA (the reference's "
                         "bitsandbytes rule; MMI_Q8_ACT=bf16 in the environment: weight-only, widened to bf16 in registers); fp8 = e4m3 linears on the fp8 MFMA")
    ap.add_argument("--kv", default="bf16", choices=["bf16", "fp8"], help="KV ring of the temporal transformer: the reference's bf16, or e4m3 (half the attention stream)")
    ap.add_argument("--launch-lists", default="", help="directory to write the step's launch lists (site per kernel launch) into, for scripts/rocpd_sites.py")
    ap.add_argument("--serial", action="store_true",
                    help="duplex workload: the reference's serving loop on ONE stream (encode -> step -> decode back to back) instead of "
                         "the three-stream pipeline of mmi_duplex_* (encode(t+1) and decode(t-1) under LMGen.step(t))")
    ap.add_argument("--lm-layers", type=int, default=0, help="debug: override the number of temporal layers (invalidates the result)")
    ap.add_argument("--kv-depth", default="mid", choices=["mid", "start", "full"],
                    help="mid (default): before the warm-up every session is moved (mmi_lm_seek) to the MIDPOINT of the configuration's run "
                         "(SURVEY.md 8d: C4 = 500 steps, sessions 8 frames apart -> ring depth 250 + 8 b; C3 = 300 steps -> depth 150), so "
                         "the measured step does not depend on --steps; start: sessions start at depth 8 b (rounds 1-3); "
                         "full: every ring full (depth = context, 3000)")
    ap.add_argument("--kv-seek", type=int, default=-1,
                    help="debug / tuning: move every session to this ring depth instead (overrides --kv-depth; not a named configuration)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra figures of the default line: `full_context` (every session 3000 positions deep), `c3` (one session), `c5` (64 sessions, int8 x int8) and `c2` (the codec alone, 8 streams)")
    ap.add_argument("--no-pin", action="store_true", help="do not pin the rank's host thread to its own block of cores")
    return ap.parse_args()


def pin_host_thread(rank, world):
    """The duplex pipeline's gate is a HOST wait (mmi_duplex_submit blocks in hipEventSynchronize), so with N ranks on one
    node N host threads' scheduling jitter lands in ms_per_step: rank r's main thread - and every thread it starts from here on
    (the HIP runtime's) - is confined to its own block of cores.  The BLAS pool numpy started at import keeps the whole machine
    (the CPU-baseline leg lifts the pin again for its own thread).  Returns the cores, for the JSON line."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None
    per = max(1, min(16, len(allowed) // max(world, 1)))
    mine = allowed[rank * per:(rank + 1) * per] or allowed
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return mine


def unpin_host_thread():
    try:
        os.sched_setaffinity(0, range(os.cpu_count() or 1))
    except (AttributeError, OSError):
        pass


def gather_per_rank(x, dist, dev, world):
    """One float per rank -> the list over ranks (every rank gets it)."""
    if dist is None:
        return [float(x)]
    t = torch.tensor([float(x)], device=dev, dtype=torch.float64)
    allr = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allr, t)
    return [float(v[0]) for v in allr]


def rank_report(rank, local, world, dist, dev):
    """One line per rank on stderr, and the same facts gathered for the JSON line: which physical device the rank drives (PCI bus
    id), what the weight broadcast at load moved and at what rate.  The first real N-GPU run is then self-diagnosing: N distinct
    bus ids = N GPUs, `bcast_world` = the ranks RCCL saw, `bcast_gb_per_s` = the rate the weights crossed xGMI at.  Gathered with
    the benchmark's one collective pattern (an all_gather of a float64 per rank, `gather_per_rank`): no object collective."""
    from moshi_amd.dist import BROADCAST_STATS as st
    try:
        pr = torch.cuda.get_device_properties(dev)
        code = (int(getattr(pr, "pci_domain_id", 0)) << 16) | (int(getattr(pr, "pci_bus_id", 0)) << 8) | int(getattr(pr, "pci_device_id", 0))
        name = pr.name
    except Exception:      # the gloo / CPU test configuration
        code, name = -1, "cpu"

    def bus(c):
        c = int(c)
        return "n/a" if c < 0 else "%04x:%02x:%02x.0" % (c >> 16, (c >> 8) & 0xff, c & 0xff)
    rate = st["bytes"] / 1e9 / st["seconds"] if st["seconds"] > 0 else None
    sys.stderr.write("bench.py rank %d/%d: %s at %s; weight broadcast over %d rank(s): %.2f GB in %.2f s, %d buckets -> %s GB/s\n"
                     % (rank, world, name, bus(code), st["world"], st["bytes"] / 1e9, st["seconds"], st["buckets"],
                        "%.2f" % rate if rate else "n/a"))
    cols = [gather_per_rank(v, dist, dev, world) for v in (code, st["world"], st["bytes"], st["seconds"], st["buckets"])]
    out = []
    for r in range(world):
        c, w, nb, sec, bk = (col[r] for col in cols)
        out.append({"rank": r, "device": name if r == rank else None, "pci_bus_id": bus(c), "bcast_world": int(w), "bcast_gb": round(nb / 1e9, 3),
                    "bcast_s": round(sec, 3), "bcast_buckets": int(bk), "bcast_gb_per_s": round(nb / 1e9 / sec, 2) if sec > 0 else None})
    return out


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args, argv=None):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: the benchmark spawns its own N ranks (one process
    per GPU, the deployment of the reference: N single-GPU replicas, swarm-config.yml:57-63) by re-executing itself under
    `torch.distributed.run` on 127.0.0.1.  Returns None when this process is already a rank (or N == 1)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    backend = os.environ.get("MMI_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible; refusing to run "
                             f"a smaller job under that label\n")
            return 2
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve())]
    cmd += list(sys.argv[1:] if argv is None else argv)
    return subprocess.call(cmd)


def dist_setup(n, backend="nccl"):
    """One process per GPU (launched by torch.distributed.run): sessions shard data-parallel, no data-path collective.
    `backend="gloo"` is the CPU stand-in used by tests/test_bench_dist.py."""
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(n, 1):
        raise SystemExit(f"bench.py: --gpus {n} but the launcher started {world} rank(s); `n_gpus` must be the ranks that ran")
    if n > 1 or world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
        return rank, local, world, dist
    if backend == "nccl":
        torch.cuda.set_device(0)
    return 0, 0, 1, None


def job_time(dt_local, dist, dev):
    """Whole-job time of the timed region = MAX over ranks (the only collective of the benchmark besides barriers)."""
    if dist is None:
        return dt_local
    t = torch.tensor([dt_local], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_latency(p50, p95, dist, dev, world):
    """BASELINE.json's second figure is the JOB's step latency: MAX over ranks, with the per-rank figures alongside."""
    if dist is None:
        return p50, p95, [p50], [p95]
    t = torch.tensor([p50, p95], device=dev, dtype=torch.float64)
    allr = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allr, t)
    p50s, p95s = [float(x[0]) for x in allr], [float(x[1]) for x in allr]
    return max(p50s), max(p95s), p50s, p95s


def job_value(world, sessions_per_gpu, steps, dt):
    """frames/s of the whole job: every rank advances `sessions_per_gpu` sessions by `steps` frames (weak scaling)."""
    return world * sessions_per_gpu * steps / dt


def mimi_algorithmic_bytes(cfg, B, frames_so_far):
    """weights once per step for the whole batch + per-stream KV/ring/IO traffic (fp32 here)."""
    from moshi_amd.weights import mimi_state_spec
    used = 0
    for name, shape, _ in mimi_state_spec(cfg):
        if "_codebook" in name:
            continue
        n = 1
        for s in shape:
            n *= s
        used += n
    used += 8 * cfg.q_bins * cfg.q_dimension           # the 8 active codebooks
    w = 4 * used
    T = cfg.resample_stride
    Lp = min(T * frames_so_far, cfg.tr_context)
    kv = 4 * 2 * cfg.tr_num_layers * 2 * cfg.tr_d_model * (Lp + T)
    rings = 2 * 4 * 24198 if cfg.dimension == 512 else 0
    io = 4 * 2 * cfg.frame_size + 8 * 8 * 2
    return w + B * (kv + rings + io)


def cpu_baseline_mimi(cfg, sd, seconds=12.0):
    """`port` baseline: the numpy oracle (oracle/mimi_oracle.py) on the host cores, B=1, bounded sample."""
    from oracle.mimi_oracle import MimiOracle
    orc = MimiOracle(sd, cfg, num_codebooks=8)
    orc.streaming(1)
    rng = np.random.default_rng(0)
    n, t0 = 0, time.perf_counter()
    while True:
        x = (0.1 * rng.standard_normal((1, 1, cfg.frame_size))).astype(np.float32)
        orc.decode(orc.encode(x))
        n += 1
        dt = time.perf_counter() - t0
        if dt > seconds or n >= 200:
            break
    return {"value": n / dt, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{n} Mimi encode+decode frames, B=1, numpy oracle (BLAS threads = host cores)"}


def served_main(args, rank, world, dist, dev, mimi, mcfg, B):
    """Every slot of a SessionBatcher holds a live channel that receives one 80 ms frame of host PCM per step; a step =
    push x B, mmi_batcher_step (H2D, encode -> LMGen.step -> decode, D2H, sync), pop x B."""
    from bench_lm import make_lm
    from moshi_amd.batcher import SessionBatcher
    lm = make_lm(dev, B, args, streaming=False)
    batcher = SessionBatcher(mimi, lm, B, seed=1234 + rank)
    rng = np.random.default_rng(1000 + rank)
    frames = (0.1 * rng.standard_normal((B, mcfg.frame_size))).astype(np.float32)
    chans = [batcher.open() for _ in range(B)]

    def step():
        for i, ch in enumerate(chans):
            batcher.push(ch, frames[i])
        n = batcher.step()
        assert n == B
        got = sum(batcher.pop(ch) is not None for ch in chans)
        return got
    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    played = 0
    for _ in range(args.steps):
        played += step()
    torch.cuda.synchronize(dev)
    dt = job_time(time.perf_counter() - t0, dist, dev)
    if dist is not None:
        dist.barrier()
    st = batcher.stats()
    out = {"metric": "12.5 Hz frames/s end-to-end Mimi+Moshi-7B, host PCM in/out through the session batcher",
           "value": job_value(world, B, args.steps, dt), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "device_ms_last_step": st["last_step_ms"],
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.quant == "none" else args.quant,
           "data": "synthetic",
           "config": {"workload": "full duplex through mmi_batcher_* (BASELINE configs[3] sessions, PCIe + host routing included)",
                      "sessions_per_gpu": B, "frames_played": played, "parallelism": f"dp{world} (independent sessions, no collective)"}}
    if rank == 0:
        print(json.dumps(out), flush=True)
    batcher.close_all()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def launchcheck_main(args, backend):
    """The launch + aggregation skeleton alone (no model): what tests/test_bench_dist.py drives through the self-launch."""
    rank, local, world, dist = dist_setup(args.gpus, backend=backend)
    dev = torch.device("cpu") if backend == "gloo" else torch.device("cuda", local)
    if dist is not None:
        dist.barrier()
    cores = None if args.no_pin else pin_host_thread(rank, world)
    dt_local = 0.5 + 0.25 * rank
    dt = job_time(dt_local, dist, dev)
    ms_ranks = [1e3 * v / args.steps for v in gather_per_rank(dt_local, dist, dev, world)]
    first_core = gather_per_rank(-1 if not cores else cores[0], dist, dev, world)
    p50, p95, p50s, p95s = job_latency(5.0 + rank, 6.0 + rank, dist, dev, world)      # rank r pretends to be r ms slower
    if dist is not None:      # the load-time collective on a small model, so that the per-rank report carries a broadcast
        from moshi_amd.config import tiny_lm_config
        from moshi_amd.dist import broadcast_state_dict
        from moshi_amd.weights import lm_state_spec, random_lm_state_dict
        tcfg = tiny_lm_config()
        broadcast_state_dict(random_lm_state_dict(tcfg, seed=1) if rank == 0 else None, lm_state_spec(tcfg), torch.bfloat16, dev, src=0)
    ranks_info = rank_report(rank, local, world, dist, dev)
    if dist is not None:
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "launchcheck", "value": job_value(world, args.batch, args.steps, dt), "unit": "frames/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
                          "p50_ms_per_step": p50, "p95_ms_per_step": p95, "p50_ms_per_rank": p50s, "p95_ms_per_rank": p95s,
                          "ms_per_step_per_rank": ms_ranks, "host_first_core_per_rank": [int(c) for c in first_core],
                          "ranks": ranks_info, "backend": backend}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    args = parse()
    rc = self_launch(args)
    if rc is not None:
        sys.exit(rc)
    if args.workload == "launchcheck":
        return launchcheck_main(args, os.environ.get("MMI_BENCH_BACKEND", "nccl"))
    rank, local, world, dist = dist_setup(args.gpus)
    dev = torch.device("cuda", local)
    cores = None if args.no_pin else pin_host_thread(rank, world)
    from moshi_amd import MimiConfig, MimiModel
    from moshi_amd.weights import random_mimi_state_dict

    t_load = time.perf_counter()
    workload = args.workload
    have_lm = (ROOT / "moshi_amd" / "lm.py").exists()
    if workload == "auto":
        workload = "duplex" if have_lm else "mimi"
    B = args.batch
    mcfg = MimiConfig()
    from bench_lm import replicated_state_dict
    from moshi_amd.weights import mimi_state_spec
    msd = replicated_state_dict(lambda: random_mimi_state_dict(mcfg, seed=1234, device=dev), mimi_state_spec(mcfg), torch.float32, dev)
    mimi = MimiModel(msd, mcfg, device=dev, max_batch=B, num_codebooks=8)
    if workload == "served":
        return served_main(args, rank, world, dist, dev, mimi, mcfg, B)
    mimi.streaming_forever(B)
    lm_gen = None
    if workload in ("duplex", "lm"):
        from bench_lm import make_lm  # noqa
        lm_gen = make_lm(dev, B, args)

    torch.cuda.synchronize(dev)
    load_local = time.perf_counter() - t_load
    load_s = job_time(load_local, dist, dev)      # rank 0 draws, RCCL broadcasts in 1 GiB buckets, every rank packs
    load_ranks = gather_per_rank(load_local, dist, dev, world)
    ranks_info = rank_report(rank, local, world, dist, dev)      # (every rank calls it: it gathers)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    pcm = 0.1 * torch.randn(B, 1, mcfg.frame_size, device=dev, generator=g)
    user_codes = torch.randint(0, mcfg.q_bins, (B, 8, 1), device=dev, generator=g)

    # the duplex step is software-pipelined by default (moshi_amd/duplex.py): frames are submitted back to back, each model keeps
    # its own stream order (bit-identical results, tests/duplex_cases.py), and the codec runs in the shadow of the LM
    dup = None
    if workload == "duplex" and not args.serial:
        from moshi_amd.duplex import DuplexStream
        dup = DuplexStream(mimi, lm_gen)

    def step():
        nonlocal user_codes
        if workload == "mimi":
            codes = mimi.encode(pcm)
            return mimi.decode(codes)
        if workload == "lm":
            return lm_gen.step(user_codes)
        if dup is not None:
            return dup.step(pcm, want_tokens=False)[1]
        codes = mimi.encode(pcm)
        tokens = lm_gen.step(codes)
        if tokens is None:
            return None
        return mimi.decode(tokens[:, 1:])      # read in place (strided); the -2 "not yet valid" rows are clamped by the decoder's gather

    def join():
        if dup is not None:
            dup.join()

    def sync():
        join()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    staggered = 0
    if lm_gen is not None:
        from bench_lm import stagger
        staggered = stagger(mimi if workload == "duplex" else None, lm_gen, step, B, args.stagger, dev, before_mask=join)
    # ring depth of session b when the warm-up starts.  "mid": the midpoint of the configuration's run as SURVEY.md 8d states it
    # (C4: 500 steps, sessions `stagger` frames apart; C3: one session, 300 steps) - the rings' skipped positions hold zeros, the
    # kernels read them all the same (no data-dependent control flow), and the codec's rings are where the stagger left them
    # (rows >= 16 of 32 are past their 250-slot wrap).  "start": what the stagger alone leaves (rounds 1-3).
    base_depth = [args.stagger * b if B > 1 else staggered for b in range(B)]
    if lm_gen is not None and args.kv_depth != "start":
        join()
        if args.kv_depth == "full":
            base_depth = [lm_gen.lm_model.config.context] * B
        else:
            base_depth = [250 + args.stagger * b for b in range(B)] if B > 1 else [150]
        if args.kv_seek >= 0:
            base_depth = [args.kv_seek] * B
        lm_gen.seek(base_depth)
    trace("staggered")
    for _ in range(args.warmup):
        step()
    sync()
    trace("warm")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    join()
    torch.cuda.synchronize(dev)
    dt_local = time.perf_counter() - t0
    dt = job_time(dt_local, dist, dev)
    ms_ranks = [1e3 * v / args.steps for v in gather_per_rank(dt_local, dist, dev, world)]
    first_core = gather_per_rank(-1 if not cores else cores[0], dist, dev, world)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)

    ms = 1e3 * dt / args.steps
    value = job_value(world, B, args.steps, dt)
    trace("timed region done")
    if args.launch_lists and rank == 0:
        d = Path(args.launch_lists)
        d.mkdir(parents=True, exist_ok=True)
        lists = {}
        if workload in ("duplex", "mimi"):
            lists["mimi_encode"], lists["mimi_decode"] = mimi.launch_list("encode"), mimi.launch_list("decode")
        if lm_gen is not None:
            lists["lm"] = lm_gen.launch_list(with_bytes=True)
        for k, v in lists.items():
            (d / f"launch_list_{k}.tsv").write_text("".join("\t".join(str(x) for x in row) + "\n" for row in v))

    # p50 / p95 latency of a single step (BASELINE.json's second figure): a separate, untimed-for-`value` pass with a
    # device event before and after every step and no host synchronisation inside the loop.  Pipelined: every frame completes
    # before the next is submitted, so the figure is PCM-in -> PCM-out of ONE frame with nothing else in flight (what a
    # session waits for), not the steady-state step interval `ms_per_step` reports.
    n_lat = max(8, min(args.steps, 40))
    join()
    torch.cuda.synchronize(dev)
    if dup is not None:
        # one frame alone through the pipeline: device timestamps from the submit reaching the caller's stream (PCM in) to the end
        # of the frame's decode (PCM out) - mmi_duplex_get_timeline - with the host waiting for the frame (mmi_duplex_flush)
        dup.timeline(True)
        lat = []
        for _ in range(n_lat):
            out = step()
            dup.flush()
            if out is not None:
                lat.append(dup.timeline()["decode"][1])
        dup.timeline(False)
        lat.sort()
    else:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_lat)]
        for a_ev, b_ev in evs:
            a_ev.record()
            step()
            b_ev.record()
        torch.cuda.synchronize(dev)
        lat = sorted(a_ev.elapsed_time(b_ev) for a_ev, b_ev in evs)
    p50, p95 = lat[len(lat) // 2], lat[min(len(lat) - 1, int(0.95 * len(lat)))]
    trace("latency pass done")
    p50, p95, p50_ranks, p95_ranks = job_latency(p50, p95, dist, dev, world)
    out = {
        "metric": "12.5 Hz frames/s end-to-end Mimi+Moshi-7B" if workload == "duplex" else f"12.5 Hz frames/s ({workload} only)",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "p50_ms_per_step": p50, "p95_ms_per_step": p95, "p50_ms_per_rank": p50_ranks, "p95_ms_per_rank": p95_ranks,
        "ms_per_step_per_rank": ms_ranks, "load_s": load_s, "load_s_per_rank": load_ranks, "ranks": ranks_info,
        "host_affinity": {"cores_per_rank": None if not cores else len(cores), "first_core_per_rank": [int(c) for c in first_core],
                          "note": "each rank's main thread (the duplex pipeline's host-kept gate) and the HIP runtime threads it starts are "
                                  "confined to a disjoint block of cores; -1 = not pinned"},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"none": "bf16", "q8": ("bf16 x int8 weights (weight-only)" if os.environ.get("MMI_Q8_ACT", "")[:1] == "b" else "int8 weights x int8 activations (int32 accumulate), bf16 elsewhere"), "fp8": "fp8 (e4m3 weights and activations, fp32 accumulate)"}[args.quant] if workload != "mimi" else "f32", "data": "synthetic",
        "config": {"workload": {"duplex": "full duplex Mimi enc -> Moshi-7B LMGen.step -> Mimi dec (BASELINE configs[3])",
                                "mimi": "Mimi streaming encode+RVQ+decode (BASELINE configs[1])",
                                "lm": "Moshi-7B LMGen.step (BASELINE configs[2])"}[workload],
                   "sessions_per_gpu": B, "parallelism": f"dp{world} (independent sessions, no collective)",
                   "pipelined": dup is not None,
                   "schedule": ("three HIP streams (encoder / LM / decoder) + events: encode(t+1) and decode(t-1) run under LMGen.step(t); "
                                "ms_per_step = steady-state interval between frames of the 32-session batch, p50/p95 = one frame alone, "
                                "PCM in -> PCM out; outputs bit-identical to the serial loop (tests/test_c_duplex_gpu.py)") if dup is not None
                               else "one stream: encode -> step -> decode back to back (server.py:132-146)",
                   "rvq": "exact fp64 argmin of ||x - e||^2 (equals the reference's fp32 cdist argmin except at fp32 near-ties: 16 of 131072 decisions, profiles/r02_logs/parity_rvq_indices_z.json)",
                   "mimi_dtype": "f32", "weights": "random-init (seeded), architecture of the named model" + {"none": "", "q8": ", LM linears row-wise int8", "fp8": ", LM linears row-wise e4m3"}[args.quant],
                   "sampling": "temp .8/.7 top-k 250/25 (LMGen defaults), on-device RNG",
                   "kv_cache": args.kv,
                   "session_stagger_frames": args.stagger if lm_gen is not None else 0,
                   "kv_depth": ({"mid": "sessions moved (mmi_lm_seek) to the midpoint of the configuration's run before the warm-up: ring depth "
                                        "250 + 8 b of SURVEY 8d C4's 500-step run (150 for the single session of C3); skipped ring rows hold zeros",
                                 "start": "sessions start at depth 8 b (the stagger alone; rounds 1-3)",
                                 "full": "every session moved (mmi_lm_seek) to position `context`: all 3000 slots of every ring are read"}[args.kv_depth]
                                if lm_gen is not None and args.kv_seek < 0 else
                                (f"--kv-seek {args.kv_seek}: every session moved to that ring depth (tuning run, not a named configuration)" if lm_gen is not None else None)),
                   "kv_positions_at_start": ([base_depth[0], base_depth[-1]] if lm_gen is not None else None),
                   "kv_positions_at_end": ([base_depth[b] + args.warmup + args.steps for b in (0, B - 1)] if lm_gen is not None else None)},
    }
    if rank == 0:
        if workload == "mimi":
            frames = args.warmup + args.steps
            nbytes = mimi_algorithmic_bytes(mcfg, B, frames)
            ach = nbytes / (ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": None,
                               "kernel": "whole Mimi step (all kernels)", "algorithmic_bytes": nbytes}
        elif lm_gen is not None:
            from bench_lm import lm_step_algorithmic_bytes, roofline_lm
            # ring depth of every session in the middle of the profiled steps (they follow the timed region and the latency pass)
            done = args.warmup + args.steps + max(8, min(args.steps, 40)) + max(4, min(args.steps, 20)) // 2
            out["roofline"] = roofline_lm(lm_gen, step, args, sync, kv_rows=[base_depth[b] + done for b in range(B)])
            # the WHOLE step against the HBM roofline: LM weights once + every session's KV at its depth at
            # the middle of the timed region (+ Mimi's weights / rings / KV when the step includes the codec)
            mid = args.warmup + args.steps // 2
            L_rows = [base_depth[b] + mid for b in range(B)]
            step_bytes = lm_step_algorithmic_bytes(lm_gen.lm_model.config, L_rows, quant=args.quant, kv=args.kv)
            parts = {"lm": step_bytes}
            if workload == "duplex":
                parts["mimi"] = mimi_algorithmic_bytes(mcfg, B, mid)
                step_bytes += parts["mimi"]
            ach = step_bytes / (ms * 1e-3) / 1e9
            out["roofline"]["step"] = {"algorithmic_bytes": step_bytes, "parts": parts, "achieved": ach, "unit": "GB/s",
                                       "frac": ach / HBM_PEAK_GBS, "ms_per_step": ms}
        if workload == "duplex" and world == 1 and not args.no_extras and not args.lm_layers and args.quant == "none" and args.kv == "bf16":
            # the other figures SURVEY 8d asks of this model, measured in the same process after the headline
            from bench_lm import extra_full_context, extra_c3
            join()
            if args.kv_depth == "mid":
                # the same pipelined step at the ring depth rounds 1-3 quoted (sessions 8 b deep + warm-up): the figure comparable
                # with BENCH_r01..r03 (6.19 ms in round 3), beside the mid-run headline
                lm_gen.seek([args.stagger * b for b in range(B)])
                for _ in range(3):
                    step()
                sync()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                join()
                torch.cuda.synchronize(dev)
                out["kv_depth_start"] = {"ms_per_step": 1e3 * (time.perf_counter() - t1) / args.steps, "steps": args.steps,
                                         "kv_positions": [0, args.stagger * (B - 1)],
                                         "note": "same pipelined step with the rings as shallow as rounds 1-3 measured them (depth 8 b); BENCH_r03: 6.19 ms"}
            out["full_context"] = extra_full_context(lm_gen, user_codes, B, dev)
            out["c3"] = extra_c3(dev, args)
            from bench_lm import extra_c5, extra_c2
            out["c5"] = extra_c5(args)
            out["c2"] = extra_c2(args)
        if not args.no_cpu_baseline and world == 1:      # the CPU leg runs at N = 1 only (the other ranks would sit in the barrier)
            unpin_host_thread()
            cpu_sd = {k: v.cpu() for k, v in msd.items()}
            base = cpu_baseline_mimi(mcfg, cpu_sd)
            if workload != "mimi":
                from bench_lm import cpu_baseline_duplex
                base = cpu_baseline_duplex(base, args, dev=dev)
            out["cpu_baseline"] = base
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
