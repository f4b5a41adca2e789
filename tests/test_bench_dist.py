"""The N > 1 path of bench.py on CPU: two processes, gloo backend, rendezvous on 127.0.0.1.  Sessions shard by rank with
no data-path collective; the only exchanges are the barrier and the MAX-reduce of the timed region (bench.py)."""
import os
import sys
from pathlib import Path

import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, out):
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    r, local, w, dist = bench.dist_setup(world, backend="gloo")
    assert (r, w) == (rank, world) and dist is not None
    dist.barrier()
    dt_local = 0.5 + 0.25 * rank                       # rank 1 is the slow one
    dt = bench.job_time(dt_local, dist, torch.device("cpu"))
    dist.barrier()
    out[rank] = (dt, bench.job_value(w, 32, 10, dt))
    dist.destroy_process_group()


def test_two_rank_timing_and_aggregate():
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, 29611, out), nprocs=world, join=True)
    assert set(out.keys()) == {0, 1}
    for rank in range(world):
        dt, value = out[rank]
        assert abs(dt - 0.75) < 1e-9                   # MAX over ranks
        assert abs(value - 2 * 32 * 10 / 0.75) < 1e-6  # whole-job frames/s: all ranks' sessions over the slowest rank's time


def test_single_rank_passthrough():
    sys.path.insert(0, str(ROOT))
    import bench
    assert bench.job_time(0.3, None, None) == 0.3
    assert bench.job_value(1, 32, 10, 0.5) == 640.0


def _bcast_worker(rank, world, port, out):
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from moshi_amd.config import tiny_lm_config
    from moshi_amd.dist import broadcast_state_dict
    from moshi_amd.weights import lm_state_spec, random_lm_state_dict
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = tiny_lm_config()
    spec = lm_state_spec(cfg)
    sd = random_lm_state_dict(cfg, seed=31) if rank == 0 else None     # only rank 0 holds the weights
    got = broadcast_state_dict(sd, spec, torch.bfloat16, "cpu", src=0, bucket_bytes=1 << 16)   # small buckets: several broadcasts
    ref = random_lm_state_dict(cfg, seed=31)
    out[rank] = (set(got) == set(ref)) and all(torch.equal(got[k], ref[k]) for k in ref)
    dist.barrier()
    dist.destroy_process_group()


def test_weight_broadcast_at_load_two_ranks():
    """The engine's one collective: rank 0's weights replicated to the other rank in flat buckets (gloo here, RCCL on GPUs)."""
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_bcast_worker, args=(world, 29613, out), nprocs=world, join=True)
    assert out[0] and out[1]


def _run_bench(extra, env_extra):
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=600)


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` (no launcher environment) starts two ranks itself; `n_gpus` is the ranks that rendezvoused."""
    import json
    r = _run_bench(["--gpus", "2", "--workload", "launchcheck", "--steps", "10"], {"MMI_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2
    assert abs(out["value"] - 2 * 32 * 10 / 0.75) < 1e-6      # both ranks' sessions over the slower rank's time
    # the latency figure is the job's: MAX over ranks (rank 1 pretends to be 1 ms slower), the per-rank figures alongside
    assert out["p50_ms_per_step"] == 6.0 and out["p95_ms_per_step"] == 7.0
    assert out["p50_ms_per_rank"] == [5.0, 6.0] and out["p95_ms_per_rank"] == [6.0, 7.0]
    # per-rank step time beside the job's (rank 0: 0.5 s, rank 1: 0.75 s over 10 steps), and each rank's host thread pinned to
    # its own block of cores
    assert out["ms_per_step_per_rank"] == [50.0, 75.0] and out["ms_per_step"] == 75.0
    cores = out["host_first_core_per_rank"]
    assert len(cores) == 2 and (cores[0] != cores[1] or len(os.sched_getaffinity(0)) < 2)
    # every rank reports the device it drives and what the weight broadcast at load moved: a first real
    # N-GPU run shows by itself that the collective saw N ranks, and at what rate
    info = out["ranks"]
    assert [r["rank"] for r in info] == [0, 1] and all(r["bcast_world"] == 2 and r["bcast_gb"] > 0 and r["bcast_buckets"] >= 1 for r in info)


def test_bench_refuses_more_gpus_than_visible():
    """On a box with fewer GPUs than requested the RCCL job must fail loudly, not print a 1-GPU line labelled N."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run_bench(["--gpus", str(have + 2), "--workload", "launchcheck"], {})
    assert r.returncode != 0
    assert "GPU(s) are visible" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_bench_rejects_rank_count_mismatch():
    """Started by a launcher with a different world size than --gpus: refuse (the line's n_gpus must be what ran)."""
    r = _run_bench(["--gpus", "4", "--workload", "launchcheck"],
                   {"MMI_BENCH_BACKEND": "gloo", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29617"})
    assert r.returncode != 0 and "rank(s)" in (r.stderr + r.stdout)
