"""The engine's one collective on a real device: `broadcast_state_dict` (moshi_amd/dist.py) packs rank 0's weights into flat
buckets, broadcasts them over RCCL and unpacks views on the receivers.  A 1-GPU box has no second rank, so the test forces the
device path at world size 1 (MMI_FORCE_BCAST): the RCCL communicator is created, every bucket goes through `ncclBroadcast`
(rank 0 -> itself), and in "recv" mode the result is what a receiving rank keeps - the views into the buckets - which must be
bit-equal to the source.  Peak device memory of the source path: weights + 2 buckets."""
import datetime
import socket

import pytest
import torch

from moshi_amd.config import LMConfig
from moshi_amd.dist import broadcast_state_dict
from moshi_amd.weights import lm_state_spec, random_lm_state_dict

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


@pytest.fixture()
def rccl_world_of_one():
    import torch.distributed as dist
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120))
    yield dist
    dist.destroy_process_group()


def test_bucketed_broadcast_runs_on_the_device_at_world_size_one(rccl_world_of_one, monkeypatch):
    cfg = LMConfig(num_layers=2)                                   # 2 temporal layers + depformer + heads: ~3 GB of bf16
    spec = lm_state_spec(cfg)
    sd = random_lm_state_dict(cfg, seed=7, device=DEV)
    weights = sum(v.numel() * v.element_size() for v in sd.values())
    bucket = 256 << 20
    n_buckets = -(-weights // bucket)
    assert n_buckets >= 8                                          # several collectives, not one
    torch.cuda.synchronize(DEV)
    # ---- the source path: its own tensors come back, the buckets are scratch; peak = weights + 2 buckets
    monkeypatch.setenv("MMI_FORCE_BCAST", "1")
    torch.cuda.reset_peak_memory_stats(DEV)
    base = torch.cuda.memory_allocated(DEV)
    out = broadcast_state_dict(sd, spec, torch.bfloat16, DEV, src=0, bucket_bytes=bucket)
    torch.cuda.synchronize(DEV)
    peak = torch.cuda.max_memory_allocated(DEV) - base
    assert all(out[k] is sd[k] for k in sd)
    assert peak <= 2 * bucket + (32 << 20), f"source path peaked {peak / 2**20:.0f} MiB above the weights; 2 buckets = {2 * bucket / 2**20:.0f} MiB"
    # ---- the receiver's unpack: views into the buckets, bit-equal to the source, one bucket set = the weights again
    monkeypatch.setenv("MMI_FORCE_BCAST", "recv")
    torch.cuda.reset_peak_memory_stats(DEV)
    base = torch.cuda.memory_allocated(DEV)
    got = broadcast_state_dict(sd, spec, torch.bfloat16, DEV, src=0, bucket_bytes=bucket)
    torch.cuda.synchronize(DEV)
    peak = torch.cuda.max_memory_allocated(DEV) - base
    assert set(got) == set(sd)
    for k, v in sd.items():
        assert got[k].data_ptr() != v.data_ptr() and got[k].shape == v.shape
        assert torch.equal(got[k].view(torch.int16), v.view(torch.int16)), k
    assert peak <= weights + 2 * bucket + (32 << 20), f"receiver path peaked at {peak / 2**20:.0f} MiB for {weights / 2**20:.0f} MiB of weights"
