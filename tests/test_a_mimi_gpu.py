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


def test_full_size_c2_recipe_crosses_the_ring_wrap(factory):
    """8 streams, 200 frames, random exec masks, one mid-run reset - the 250-slot rings of the codec's
    transformers wrap at frame 126 (the state bench.py runs in, never parity-tested at full size before)."""
    import json
    from pathlib import Path
    # the engine runs the 8 streams; the numpy checker follows three of them (the always-executing row whose rings wrap at frame
    # 126, the nine-in-ten row (wraps ~140), the sine row, which is also reset mid-run) for 160 frames: ~85 s instead of 210 in a
    # suite with a time budget (four rows: 114 s on MI355X in round 5)
    res = mimi_cases.check_c2_recipe(factory, DEV, MimiConfig(), B=8, F=160, oracle_rows=[0, 1, 7])
    assert res["wrapped"]
    out = Path(__file__).resolve().parent.parent / "gpurun_out"
    if out.is_dir():
        (out / "parity_mimi_c2_recipe.json").write_text(json.dumps(res, indent=1))


def test_rvq_indices_bit_exact_on_many_vectors(factory):
    """131 072 index decisions (4096 latents x 32 levels) against the oracle's fp32 cdist restatement.
    The engine evaluates distances in fp64, i.e. it returns the exact nearest centroid; the fp32 reference formula
    can only disagree with that at a near-tie below its own resolution (relative gaps down to
    3e-6 occur about once in 1e5 decisions).  Any disagreement is therefore audited in fp64: it must be such a
    near-tie AND the engine must hold the exact argmin."""
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
    bad = np.argwhere((ce != co).any(1))
    # the observed count goes on record: printed, and written next to the other parity summaries
    n_dec = int((ce != co).sum())
    print(f"[parity] rvq_indices: {len(bad)} of 4096 vectors ({n_dec} of {ce.size} index decisions) differ from the fp32 cdist restatement")
    import json
    from pathlib import Path
    out = Path(__file__).resolve().parent.parent / "gpurun_out"
    if out.is_dir():
        (out / "parity_rvq_indices.json").write_text(json.dumps({"case": "rvq_indices", "vectors": 4096, "vectors_differing": len(bad),
                                                                  "decisions": int(ce.size), "decisions_differing": n_dec}, indent=1))
    assert len(bad) <= 4, f"{len(bad)} of 4096 vectors differ from the oracle - more than near-ties can explain"
    for b, t in bad:
        k = int(np.argmax(ce[b, :, t] != co[b, :, t]))          # first level that differs; the prefix is identical
        part = 0 if k < cfg.q_n_q_semantic else 1
        x = (orc.in_proj[part].astype(np.float64) @ lat[b, :, t].astype(np.float64)).astype(np.float32)
        k0 = 0 if part == 0 else cfg.q_n_q_semantic
        for kk in range(k0, k):
            x = (x - orc.codebooks[kk][co[b, kk, t]]).astype(np.float32)
        E = orc.codebooks[k].astype(np.float64)
        d = ((E - x.astype(np.float64)) ** 2).sum(-1)
        de, do = d[ce[b, k, t]], d[co[b, k, t]]
        assert abs(de - do) <= 2e-5 * min(de, do), f"not a near-tie: {de} vs {do}"
        assert de <= do, "the engine must hold the exact (fp64) nearest centroid"


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


@pytest.mark.parametrize("B", [8, 32])
def test_codec_is_bit_reproducible_between_streams(factory, B):
    """The full-size codec at the C2 batch (8 streams) and the benchmark batch (32): three frames encoded and decoded, then four
    more streams on the SAME handle fed the same PCM - codes and PCM bit for bit equal to the first stream's (a launch that is not
    a function of its inputs - round 4's LM failure - is the one bug class a tolerance against the checker cannot see)."""
    cfg = MimiConfig()
    sd = random_mimi_state_dict(cfg, seed=1234)
    g = torch.Generator().manual_seed(19 + B)
    x = (0.25 * torch.randn(B, 1, cfg.frame_size * 3, generator=g)).to(DEV)
    m = factory(sd, cfg, 8, max_batch=B)

    def run():
        out = []
        with m.streaming(B):
            for f in range(3):
                codes = m.encode(x[:, :, f * cfg.frame_size:(f + 1) * cfg.frame_size])
                out.append((codes.clone(), m.decode(codes).clone()))
        return out
    first = run()
    for r in range(4):
        for f, ((c0, p0), (c1, p1)) in enumerate(zip(first, run())):
            assert torch.equal(c0, c1), f"repeat {r} frame {f}: codes differ between two streams fed the same PCM"
            assert torch.equal(p0, p1), f"repeat {r} frame {f}: PCM differs between two streams fed the same PCM"


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
