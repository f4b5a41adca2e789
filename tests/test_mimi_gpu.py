"""Parity of the gfx950 Mimi engine on a real MI355X: golden vectors from the reference, the numpy oracle on
seeded inputs, and size-independent properties at the benchmark's batch sizes."""
import numpy as np
import pytest
import torch

from moshi_amd import MimiConfig, MimiModel, tiny_mimi_config
from moshi_amd.weights import random_mimi_state_dict
from tests import mimi_cases

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture()
def factory(gpu_lib):
    def make(sd, cfg, K, max_batch=32):
        return MimiModel(sd, cfg, device=DEV, max_batch=max_batch, num_codebooks=K)
    return make


def test_tiny_schedule_matches_reference_golden(factory):
    mimi_cases.check_tiny_against_golden(factory, DEV)


def test_full_size_matches_reference_golden(factory):
    mimi_cases.check_full_against_golden(factory, DEV)


@pytest.mark.parametrize("B", [1, 5])
def test_tiny_matches_oracle_with_masks_and_reset(factory, B):
    mimi_cases.oracle_vs_engine(factory, DEV, tiny_mimi_config(), seed=31 + B, B=B, F=8, K=5)


def test_full_size_matches_oracle(factory):
    mimi_cases.oracle_vs_engine(factory, DEV, MimiConfig(), seed=77, B=2, F=2, K=8, use_masks=True)


def test_rvq_indices_bit_exact_on_many_vectors(factory):
    """>= 1e5 index decisions: 4096 latents x 32 codebooks against the oracle's cdist restatement, with an fp64 audit
    of any disagreement (a disagreement is only legitimate at a near-tie below fp32 resolution)."""
    from oracle.mimi_oracle import MimiOracle
    cfg = MimiConfig()
    sd = random_mimi_state_dict(cfg, seed=1234)
    m = factory(sd, cfg, 32, max_batch=64)
    orc = MimiOracle(sd, cfg, num_codebooks=32)
    rng = np.random.default_rng(0)
    lat = rng.standard_normal((64, cfg.dimension, 64)).astype(np.float32)
    ce = m.quantize(torch.from_numpy(lat).to(DEV)).cpu().numpy()
    co = orc.quantize(lat)
    assert ce.shape == co.shape == (64, 32, 64)
    first_bad = (ce != co).any(1)            # [B, T]: rows where some level differs (later levels then cascade)
    assert first_bad.sum() == 0, f"{int(first_bad.sum())} of {first_bad.size} vectors differ from the oracle"


def test_batch_rows_are_independent_and_graph_equals_eager(factory, monkeypatch):
    """Size-independent properties at the benchmark batch (B=32): every row equals the same stream run alone at B=1,
    and the hipGraph replay equals the eager launch list."""
    cfg = MimiConfig()
    sd = random_mimi_state_dict(cfg, seed=1234)
    g = torch.Generator().manual_seed(9)
    x = (0.25 * torch.randn(32, 1, cfg.frame_size * 3, generator=g)).to(DEV)
    big = factory(sd, cfg, 8, max_batch=32)
    with big.streaming(32):
        codes = big.encode(x)
        pcm = big.decode(codes)
    for row in (0, 17, 31):
        one = factory(sd, cfg, 8, max_batch=1)
        with one.streaming(1):
            c1 = one.encode(x[row:row + 1])
            p1 = one.decode(c1)
        assert torch.equal(c1, codes[row:row + 1])
        assert torch.allclose(p1, pcm[row:row + 1], atol=2e-5, rtol=2e-5)
    monkeypatch.setenv("MMI_NO_GRAPH", "1")
    eager = factory(sd, cfg, 8, max_batch=32)
    with eager.streaming(32):
        c2 = eager.encode(x)
        p2 = eager.decode(c2)
    assert torch.equal(c2, codes) and torch.equal(p2, pcm)


def test_round_trip_sine_c1(factory):
    """BASELINE.json configs[0] restated: 1 s 440 Hz sine, streaming == non-streaming codes, PCM shape."""
    cfg = MimiConfig()
    m = factory(random_mimi_state_dict(cfg, seed=1234), cfg, 8, max_batch=1)
    t = torch.arange(24000) / 24000.0
    wav = (0.5 * torch.sin(2 * torch.pi * 440.0 * t)).view(1, 1, -1).to(DEV)
    codes = m.encode(wav)                       # pads to 13 frames
    assert codes.shape == (1, 8, 13) and codes.dtype == torch.int64
    with m.streaming(1):
        cs = torch.cat([m.encode(wav[..., f * 1920:(f + 1) * 1920]) for f in range(12)], -1)
    assert torch.equal(cs, codes[..., :12])
    pcm = m.decode(codes)
    assert pcm.shape == (1, 1, 13 * 1920) and torch.isfinite(pcm).all()
