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
