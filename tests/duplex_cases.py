"""Cases for the duplex pipeline (moshi_amd/csrc/duplex.hip) shared by the simulator tests (CPU) and the GPU tests.

The pipeline must not change a single bit: `DuplexStream.step` is compared with the reference's serving loop
(server.py:132-146: encode -> step -> decode on one stream) on the same inputs, sampled (on-device RNG, same seed) and greedy,
including an exec-mask change and a partial reset issued between frames after a `join()`.
"""
from __future__ import annotations

import numpy as np
import torch

from moshi_amd.duplex import DuplexStream
from moshi_amd.lm import LMGen
from tests.batcher_cases import tiny_pair


def _serial(mimi, gen, frames, events, dev):
    out = []
    for t, x in enumerate(frames):
        for kind, mask in events.get(t, []):
            m = torch.from_numpy(mask).to(dev)
            if kind == "reset":
                mimi.reset_streaming(m); gen.reset_streaming(m)
            else:
                mimi.set_exec_mask(m); gen.set_exec_mask(m)
        codes = mimi.encode(torch.from_numpy(x).to(dev))
        tokens = gen.step(codes)
        if tokens is None:
            out.append(None)
            continue
        pcm = mimi.decode(tokens[:, 1:])
        out.append((tokens.cpu().numpy().copy(), pcm.cpu().numpy().copy()))
    return out


def _pipelined(mimi, gen, frames, events, dev, join_every):
    dup = DuplexStream(mimi, gen, depth=max(4, join_every + 1))
    out, pending = [], []

    def drain():
        dup.join()
        for tok, pcm in pending:
            out.append(None if tok is None else (tok.cpu().numpy().copy(), pcm.cpu().numpy().copy()))
        pending.clear()
    for t, x in enumerate(frames):
        if t in events:
            drain()                         # mask / reset calls go through the handles on the caller's stream, after a join
            for kind, mask in events[t]:
                m = torch.from_numpy(mask).to(dev)
                if kind == "reset":
                    mimi.reset_streaming(m); gen.reset_streaming(m)
                else:
                    mimi.set_exec_mask(m); gen.set_exec_mask(m)
        pending.append(dup.step(torch.from_numpy(x).to(dev)))
        if len(pending) >= join_every:
            drain()
    drain()
    dup.close()
    return out


def check_pipeline_is_bit_identical(device, lib, B=3, steps=9, use_sampling=True, join_every=3, pair=None):
    mimi, lm, mcfg, lcfg = pair or tiny_pair(device, lib, B)
    dev = torch.device(device)
    rng = np.random.default_rng(7)
    frames = [(0.1 * rng.standard_normal((B, 1, mcfg.frame_size))).astype(np.float32) for _ in range(steps)]
    ones = np.ones(B, bool)
    m1 = ones.copy(); m1[B - 1] = False
    r1 = np.zeros(B, bool); r1[0] = True
    events = {4: [("mask", m1)], 6: [("mask", ones), ("reset", r1)]} if B > 1 else {}
    runs = []
    for fn in (_serial, lambda *a: _pipelined(*a, join_every)):
        gen = LMGen(lm, use_sampling=use_sampling, temp=0.8, temp_text=0.7, top_k=5, top_k_text=5, seed=99)
        with mimi.streaming(B), gen.streaming(B):
            runs.append(fn(mimi, gen, frames, events, dev))
    a, b = runs
    assert len(a) == len(b) == steps
    n_valid = 0
    for t, (x, y) in enumerate(zip(a, b)):
        assert (x is None) == (y is None), f"frame {t}: None pattern differs"
        if x is None:
            continue
        n_valid += 1
        assert np.array_equal(x[0], y[0]), f"frame {t}: tokens differ"
        assert np.array_equal(x[1].view(np.uint32), y[1].view(np.uint32)), f"frame {t}: PCM differs"
    assert n_valid >= steps - 2
    return n_valid


def check_strided_decode(device, lib, B=2):
    """`mimi.decode(tokens[:, 1:])` reads the column slice in place and equals the decode of a contiguous copy."""
    mimi, lm, mcfg, lcfg = tiny_pair(device, lib, B)
    dev = torch.device(device)
    g = torch.Generator().manual_seed(3)
    tokens = torch.randint(0, lcfg.card, (B, 1 + lcfg.dep_q, 1), generator=g).to(dev)
    with mimi.streaming(B):
        a = mimi.decode(tokens[:, 1:]).cpu().numpy()
    with mimi.streaming(B):
        b = mimi.decode(tokens[:, 1:].contiguous()).cpu().numpy()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
