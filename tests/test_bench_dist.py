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
