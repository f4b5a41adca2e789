"""Cases for the session batcher (moshi_amd/csrc/batcher.hip) shared by the simulator tests (CPU) and the GPU tests.

What is pinned here is the model loop of SURVEY.md 8f-1 (rust/moshi-server/src/batched_asr.rs:188-437; server.py:132-146,
163-164): which rows run, which are reset, what reaches which channel.  The arithmetic of the rows themselves is pinned
against the oracle / the reference golden vectors by the Mimi and LM cases; here the batcher is compared bit for bit with
the same schedule driven by hand through `MimiModel` / `LMGen` (the reference's own API).
"""
from __future__ import annotations

from dataclasses import replace

import numpy as np
import torch

from moshi_amd.batcher import SessionBatcher
from moshi_amd.config import tiny_lm_config, tiny_mimi_config
from moshi_amd.lm import LMGen, LMModel
from moshi_amd.mimi import MimiModel
from moshi_amd.weights import random_lm_state_dict, random_mimi_state_dict


def tiny_pair(device, lib, slots, seed=5, lm_rows=None, fuser=None):
    lcfg = tiny_lm_config()
    mcfg = replace(tiny_mimi_config(), q_bins=lcfg.card, q_n_q=lcfg.dep_q)
    msd = random_mimi_state_dict(mcfg, seed=seed)
    lsd = random_lm_state_dict(lcfg, seed=seed + 1)
    mimi = MimiModel(msd, mcfg, device=device, max_batch=slots, num_codebooks=lcfg.dep_q, lib=lib)
    lm = LMModel(lsd, lcfg, device=device, max_batch=lm_rows or slots, lib=lib, fuser=fuser)
    return mimi, lm, mcfg, lcfg


class ManualLoop:
    """The batcher's schedule written against the reference's Python API, row by row."""

    def __init__(self, mimi: MimiModel, lm: LMModel, slots: int, reset_codec_after_first_frame=True, **gen_kwargs):
        self.mimi, self.lm = mimi, lm
        self.gen = LMGen(lm, use_sampling=False, support_out_of_sync=True, **gen_kwargs)
        self.B = slots
        self.first_quirk = reset_codec_after_first_frame
        mimi.streaming_forever(slots)                       # server.py:59-60
        self.gen.streaming_forever(slots)
        self.dev = mimi.device

    def stop(self):
        self.mimi._stop_streaming()
        self.gen._stop_streaming()

    def step(self, frames, resets, firsts):
        """frames: {row: pcm[F]}; resets / firsts: sets of rows.  Returns {row: (pcm, tokens)} for rows that played."""
        B, F, dev = self.B, self.mimi.frame_size, self.dev

        def mask(rows):
            m = torch.zeros(B, dtype=torch.bool)
            for r in rows:
                m[r] = True
            return m.to(dev)

        if resets:                                          # server.py:163-164
            self.mimi.reset_streaming(mask(resets))
            self.gen.reset_streaming(mask(resets))
        if not frames:
            return {}
        pcm = torch.zeros(B, 1, F)
        for r, x in frames.items():
            pcm[r, 0] = torch.from_numpy(x)
        ex = mask(frames.keys())
        self.mimi.set_exec_mask(ex)
        self.gen.set_exec_mask(ex)
        codes = self.mimi.encode(pcm.to(dev))               # server.py:132
        if firsts and self.first_quirk:
            self.mimi.reset_streaming(mask(firsts))         # server.py:135-141
        tokens = self.gen.step(codes)                       # [B, 1 + dep_q, 1], -2 where not generated yet
        played = ex & (tokens[:, :, 0] >= 0).all(dim=1)
        self.mimi.set_exec_mask(played)
        out = self.mimi.decode(tokens[:, 1:].clamp(min=0))  # server.py:81
        tok, out, played = tokens[:, :, 0].cpu().numpy(), out.cpu().numpy(), played.cpu().numpy()
        return {r: (out[r, 0], tok[r]) for r in range(B) if played[r]}


# (step, action, session): sessions join and leave while others are mid-dialogue; "skip" = no audio arrived this step
SCRIPT = [
    (0, "open", "x"), (2, "open", "y"), (3, "skip", "x"), (5, "close", "x"), (6, "open", "z"), (6, "skip", "y"),
    (7, "open", "w"), (8, "close", "y"),
]
N_STEPS = 10


def scripted_run(batcher: SessionBatcher, manual: ManualLoop | None, frame_size: int, script=SCRIPT, n_steps=N_STEPS, seed=0):
    """Drives `batcher` (and, in lock step, `manual`) through the script; returns {session: [(pcm, tokens), ...]}."""
    rng = {}
    live, slot_of, frames_fed, chan = [], {}, {}, {}
    results, results_manual = {}, {}
    for step in range(n_steps):
        acts = [(a, s) for (t, a, s) in script if t == step]
        resets, skips = set(), set()
        for a, s in acts:
            if a == "open":
                chan[s] = batcher.open()
                used = set(slot_of.values())
                slot_of[s] = min(r for r in range(batcher.total_slots) if r not in used)   # the batcher takes the lowest free slot
                live.append(s)
                frames_fed[s] = 0
                rng[s] = np.random.default_rng(seed + sum(map(ord, s)))
                results[s], results_manual[s] = [], []
                resets.add(slot_of[s])
            elif a == "close":
                batcher.close(chan[s])
                live.remove(s)
                del slot_of[s]
            elif a == "skip":
                skips.add(s)
        frames, firsts = {}, set()
        for s in live:
            if s in skips:
                continue
            x = (0.3 * rng[s].standard_normal(frame_size)).astype(np.float32)
            # audio arrives in odd-sized pieces; the batcher cuts frames itself (batched_asr.rs:77-90)
            cut = int(rng[s].integers(1, frame_size))
            batcher.push(chan[s], x[:cut])
            batcher.push(chan[s], x[cut:])
            frames[slot_of[s]] = x
            if frames_fed[s] == 0:
                firsts.add(slot_of[s])
            frames_fed[s] += 1
        n = batcher.step()
        assert n == len(frames)
        for s in live:
            while True:
                f = batcher.pop(chan[s])
                if f is None:
                    break
                results[s].append(f)
        if manual is not None:
            got = manual.step(frames, resets, firsts)
            for s in live:
                if slot_of[s] in got:
                    results_manual[s].append(got[slot_of[s]])
    return results, results_manual


def check_batcher_matches_manual_api(device, lib):
    slots = 3
    mimi_a, lm_a, mcfg, lcfg = tiny_pair(device, lib, slots)
    mimi_b, lm_b, _, _ = tiny_pair(device, lib, slots)
    manual = ManualLoop(mimi_b, lm_b, slots)
    with SessionBatcher(mimi_a, lm_a, slots, use_sampling=False) as batcher:
        assert batcher.total_slots == slots and batcher.used_slots == 0
        res, ref = scripted_run(batcher, manual, mcfg.frame_size)
        st = batcher.stats()
    manual.stop()
    assert st["steps"] == N_STEPS and st["dropped_frames"] == 0
    played = 0
    for s in res:
        assert len(res[s]) == len(ref[s]), f"session {s}: {len(res[s])} frames from the batcher, {len(ref[s])} by hand"
        for i, ((pa, ta), (pb, tb)) in enumerate(zip(res[s], ref[s])):
            assert np.array_equal(ta, tb), f"session {s} frame {i}: tokens differ"
            assert np.array_equal(pa, pb), f"session {s} frame {i}: PCM differs"
            assert (ta >= 0).all() and ta[0] < lcfg.text_card and (ta[1:] < lcfg.card).all()
        played += len(res[s])
    # every session loses exactly max_delay frames to the LM's delay ring (lm.py:774-782), skipped steps produce nothing
    assert len(res["x"]) == 5 - 1 - lcfg.max_delay and played > 8


def check_session_independent_of_neighbours(device, lib):
    """A session's tokens and audio do not depend on who else is in the batch (rows never mix; greedy sampling)."""
    slots = 3
    mimi, lm, mcfg, _ = tiny_pair(device, lib, slots)
    alone = [(0, "open", "x"), (3, "skip", "x")]
    with SessionBatcher(mimi, lm, slots, use_sampling=False) as b:
        r_alone, _ = scripted_run(b, None, mcfg.frame_size, script=alone, n_steps=7)
    crowd = [(0, "open", "x"), (3, "skip", "x"), (1, "open", "y"), (2, "open", "z"), (4, "close", "y"), (5, "open", "w"),
             (6, "skip", "z")]
    with SessionBatcher(mimi, lm, slots, use_sampling=False) as b:
        r_crowd, _ = scripted_run(b, None, mcfg.frame_size, script=crowd, n_steps=7)
    assert len(r_alone["x"]) == len(r_crowd["x"]) > 3
    for (pa, ta), (pb, tb) in zip(r_alone["x"], r_crowd["x"]):
        assert np.array_equal(ta, tb) and np.array_equal(pa, pb)


def check_slots_and_buffers(device, lib):
    slots = 2
    mimi, lm, mcfg, _ = tiny_pair(device, lib, slots)
    import pytest
    with SessionBatcher(mimi, lm, slots, use_sampling=False, max_buffered_frames=3) as b:
        a, c = b.open(), b.open()
        assert a != c and b.used_slots == 2
        with pytest.raises(BufferError):                    # py_module.rs:443-470: no free slot
            b.open()
        b.close(a)
        assert b.used_slots == 1
        d = b.open()                                        # the freed slot is reused under a new channel id
        assert d not in (a, c)
        with pytest.raises(ValueError):
            b.push(a, np.zeros(4, np.float32))              # a closed channel id is gone for good
        assert b.step() == 0 and b.pop(c) is None           # nothing buffered -> nothing runs
        b.push(c, np.zeros(3 * mcfg.frame_size, np.float32))
        with pytest.raises(BufferError):                    # input FIFO capped at max_buffered_frames
            b.push(c, np.zeros(1, np.float32))
        for _ in range(3):
            assert b.step() == 1                            # one frame per channel per step, like the reference loop
        assert b.step() == 0
        assert b.stats()["frames"] == 3


def check_batcher_with_guidance(device, lib):
    """Guided sessions (cfg_coef 2, masked-until, one shared `sum` condition: two model rows per slot) through the batcher ==
    the same schedule by hand through a guided LMGen."""
    from moshi_amd.lm import ConditionFuser
    slots = 2
    lcfg = tiny_lm_config()
    rng = np.random.default_rng(3)
    cond = torch.from_numpy(0.5 * rng.standard_normal((2 * slots, 1, lcfg.dim)).astype(np.float32)).to(torch.bfloat16)
    kw = dict(cfg_coef=2.0, cfg_is_masked_until=[1, 2], condition_tensors={"c": (cond, None)})
    fuser = ConditionFuser({"sum": ["c"]})
    mimi_a, lm_a, mcfg, _ = tiny_pair(device, lib, slots, lm_rows=2 * slots, fuser=fuser)
    mimi_b, lm_b, _, _ = tiny_pair(device, lib, slots, lm_rows=2 * slots, fuser=fuser)
    manual = ManualLoop(mimi_b, lm_b, slots, **kw)
    script = [(0, "open", "x"), (1, "open", "y"), (3, "skip", "x"), (4, "close", "y"), (4, "open", "z")]
    with SessionBatcher(mimi_a, lm_a, slots, use_sampling=False, **kw) as batcher:
        res, ref = scripted_run(batcher, manual, mcfg.frame_size, script=script, n_steps=7)
    manual.stop()
    assert sum(len(v) for v in res.values()) > 6
    for s in res:
        assert len(res[s]) == len(ref[s])
        for (pa, ta), (pb, tb) in zip(res[s], ref[s]):
            assert np.array_equal(ta, tb) and np.array_equal(pa, pb)


def check_asr_batcher(device, lib):
    """The reference's batched ASR server proper (batched_asr.rs): channels of PCM in, text tokens out, no decoder; against the
    same schedule by hand through MimiModel.encode + LMGen.step."""
    from moshi_amd.config import tiny_stt_config
    slots = 2
    lcfg = tiny_stt_config()
    mcfg = replace(tiny_mimi_config(), q_bins=lcfg.card, q_n_q=lcfg.n_q)

    def pair():
        mimi = MimiModel(random_mimi_state_dict(mcfg, seed=5), mcfg, device=device, max_batch=slots, num_codebooks=lcfg.n_q, lib=lib)
        return mimi, LMModel(random_lm_state_dict(lcfg, seed=6), lcfg, device=device, max_batch=slots, lib=lib)
    mimi_a, lm_a = pair()
    mimi_b, lm_b = pair()
    gen = LMGen(lm_b, use_sampling=False, support_out_of_sync=True)
    mimi_b.streaming_forever(slots); gen.streaming_forever(slots)
    rng = np.random.default_rng(9)
    F = mcfg.frame_size
    with SessionBatcher(mimi_a, lm_a, slots, use_sampling=False, reset_codec_after_first_frame=False) as b:
        chans = [b.open(), b.open()]
        got = [[], []]
        ref = [[], []]
        for s in range(7):
            x = (0.3 * rng.standard_normal((slots, F))).astype(np.float32)
            active = [True, s != 3]                                  # channel 1 has no audio at step 3
            for i, ch in enumerate(chans):
                if active[i]:
                    b.push(ch, x[i])
            assert b.step() == sum(active)
            for i, ch in enumerate(chans):
                while (fr := b.pop(ch)) is not None:
                    got[i].append(int(fr[1][0]))
                    assert fr[1].shape == (1,)
            ex = torch.tensor(active, device=device)
            mimi_b.set_exec_mask(ex); gen.set_exec_mask(ex)
            pcm = torch.from_numpy(np.where(np.array(active)[:, None], x, 0.0).astype(np.float32))[:, None].to(device)
            tokens = gen.step(mimi_b.encode(pcm))
            for i in range(slots):
                if active[i] and int(tokens[i, 0, 0]) >= 0:
                    ref[i].append(int(tokens[i, 0, 0]))
    mimi_b._stop_streaming(); gen._stop_streaming()
    assert got == ref and len(got[0]) == 7 - lcfg.max_delay and len(got[1]) == 6 - lcfg.max_delay
