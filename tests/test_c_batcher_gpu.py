"""Session batcher on a real MI355X: the same schedule cases as on the simulator, through hipGraph replay,
pinned staging and the batcher's own stream, plus a full-size Mimi + Moshi-width LM run."""
import numpy as np
import pytest
import torch

from tests import batcher_cases

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_batcher_matches_the_schedule_driven_by_hand(gpu_lib):
    batcher_cases.check_batcher_matches_manual_api(DEV, None)


def test_session_is_independent_of_its_neighbours(gpu_lib):
    batcher_cases.check_session_independent_of_neighbours(DEV, None)


def test_slots_and_buffer_limits(gpu_lib):
    batcher_cases.check_slots_and_buffers(DEV, None)


def test_guided_sessions_through_the_batcher(gpu_lib):
    batcher_cases.check_batcher_with_guidance(DEV, None)


def test_batched_asr_text_only(gpu_lib):
    batcher_cases.check_asr_batcher(DEV, None)


def test_full_size_codec_with_moshi_width_lm(gpu_lib):
    """Real Mimi (24 kHz, 1920-sample frames, 2048-entry codebooks) + an LM at Moshi-7B's widths (2 temporal layers):
    18 slots, channels joining every other step; every played frame is finite audio and in-range tokens, and a channel's
    stream is identical whether it shares the batch or not."""
    from moshi_amd import MimiConfig, MimiModel
    from moshi_amd.batcher import SessionBatcher
    from moshi_amd.config import LMConfig
    from moshi_amd.lm import LMModel
    from moshi_amd.weights import random_lm_state_dict, random_mimi_state_dict
    mcfg, lcfg = MimiConfig(), LMConfig(num_layers=2, context=64)
    slots = 18
    mimi = MimiModel(random_mimi_state_dict(mcfg, seed=3, device=DEV), mcfg, device=DEV, max_batch=slots, num_codebooks=8)
    lm = LMModel(random_lm_state_dict(lcfg, seed=4), lcfg, device=DEV, max_batch=slots)
    F = mcfg.frame_size

    def run(n_channels, steps=8):
        rngs = [np.random.default_rng(100 + i) for i in range(n_channels)]
        got = [[] for _ in range(n_channels)]
        with SessionBatcher(mimi, lm, slots, use_sampling=False) as b:
            chans = []
            for s in range(steps):
                if len(chans) < n_channels and s % 2 == 0:
                    for _ in range(min(n_channels - len(chans), 6 if s else 1)):
                        chans.append(b.open())
                for i, ch in enumerate(chans):
                    b.push(ch, (0.1 * rngs[i].standard_normal(F)).astype(np.float32))
                assert b.step() == len(chans)
                for i, ch in enumerate(chans):
                    f = b.pop(ch)
                    if f is not None:
                        got[i].append(f)
            st = b.stats()
        return got, st

    crowd, st = run(13)
    assert st["last_step_ms"] > 0 and st["dropped_frames"] == 0
    assert len(crowd[0]) == 8 - lcfg.max_delay
    for frames in crowd:
        for pcm, tok in frames:
            assert np.isfinite(pcm).all() and 0 <= tok[0] < lcfg.text_card and ((tok[1:] >= 0) & (tok[1:] < lcfg.card)).all()
    alone, _ = run(1)
    assert len(alone[0]) == len(crowd[0])
    for (pa, ta), (pb, tb) in zip(alone[0], crowd[0]):
        assert np.array_equal(ta, tb) and np.array_equal(pa, pb)
