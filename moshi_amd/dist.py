"""Weight replication at load for the one-process-per-GPU deployment: rank `src` holds the state dict
(read from disk, or drawn), every other rank receives it through `torch.distributed` - backend "nccl" is RCCL over xGMI on
MI355X nodes, "gloo" in the CPU tests.  This is the ONLY collective of the engine: the frame step itself shards by
session and never communicates.

Tensors are packed into a few large flat buckets per dtype (xGMI rings are per-link bound, so fewer, larger broadcasts:
the 14.75 GB of Moshi-7B go out as 1 GiB pieces) instead of one collective per tensor.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

Spec = Sequence[Tuple[str, Tuple[int, ...], str]]

# What the replication cost on THIS rank, accumulated over every broadcast_state_dict call of the process: bytes that went through
# the collective, seconds inside it (device-synchronised around each bucket), buckets.  bench.py prints it per rank, next to the
# rank's PCI bus id, so that the first real multi-GPU run says by itself whether RCCL saw N ranks and at what rate the weights
# crossed xGMI.
BROADCAST_STATS = {"bytes": 0, "seconds": 0.0, "buckets": 0, "world": 1}


def _numel(shape) -> int:
    n = 1
    for s in shape:
        n *= int(s)
    return n


def broadcast_state_dict(state_dict: Optional[Dict[str, torch.Tensor]], spec: Spec, dtype: torch.dtype, device,
                         src: int = 0, bucket_bytes: int = 1 << 30, group=None) -> Dict[str, torch.Tensor]:
    """Replicate rank `src`'s `state_dict` (keys / shapes as in `spec`, all of `dtype`) to every rank.

    `spec` is what `moshi_amd.weights.lm_state_spec` / `mimi_state_spec` return, so that the receiving ranks can allocate
    without any metadata exchange.  Returns the full state dict on every rank (on `device`)."""
    import os
    import torch.distributed as dist
    # MMI_FORCE_BCAST (test switch; tests/test_c_loaders_gpu.py): "1" = run the bucket pack + collective even when the job has one
    # rank (so the device path executes on a 1-GPU box); "recv" = this rank also UNPACKS like a receiver (its result is the
    # views into the buckets, not its own tensors)
    force = os.environ.get("MMI_FORCE_BCAST", "")
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        assert state_dict is not None
        return state_dict
    rank = dist.get_rank(group)
    as_receiver = force == "recv"
    esize = torch.empty((), dtype=dtype).element_size()
    per_bucket = max(1, bucket_bytes // esize)
    out: Dict[str, torch.Tensor] = {}
    # greedy bucketing in spec order (identical on every rank)
    buckets: List[List[Tuple[str, Tuple[int, ...]]]] = [[]]
    fill = 0
    for name, shape, _ in spec:
        n = _numel(shape)
        if fill and fill + n > per_bucket:
            buckets.append([])
            fill = 0
        buckets[-1].append((name, tuple(shape)))
        fill += n
    for items in buckets:
        total = sum(_numel(s) for _, s in items)
        flat = torch.empty(total, dtype=dtype, device=device)
        if rank == src:
            assert state_dict is not None, "the source rank must hold the weights"
            at = 0
            for name, shape in items:
                n = _numel(shape)
                flat[at:at + n].copy_(state_dict[name].reshape(-1).to(device=device, dtype=dtype))
                at += n
        import time
        on_gpu = torch.device(device).type == "cuda"
        if on_gpu:
            torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        dist.broadcast(flat, src=src, group=group)
        if on_gpu:
            torch.cuda.synchronize(device)
        BROADCAST_STATS["seconds"] += time.perf_counter() - t0
        BROADCAST_STATS["bytes"] += total * esize
        BROADCAST_STATS["buckets"] += 1
        BROADCAST_STATS["world"] = dist.get_world_size(group)
        at = 0
        for name, shape in items:
            n = _numel(shape)
            # the source keeps its own tensors (the bucket is scratch there); receivers keep views into the bucket
            out[name] = state_dict[name] if (rank == src and not as_receiver) else flat[at:at + n].view(shape)
            at += n
    return out
