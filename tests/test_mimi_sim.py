"""Mimi engine kernels run on the CPU kernel simulator (tests/hipsim) against the golden vectors and the oracle.
This exercises the exact kernel sources the gfx950 build compiles (indexing, ring state, masks, packing); the
numerics on real MFMA hardware are covered by the -m gpu tests."""
import numpy as np
import pytest
import torch

from moshi_amd import MimiModel, tiny_mimi_config
from moshi_amd.weights import random_mimi_state_dict
from tests import mimi_cases


@pytest.fixture()
def factory(sim_lib):
    def make(sd, cfg, K, max_batch=8):
        return MimiModel(sd, cfg, device="cpu", max_batch=max_batch, num_codebooks=K, lib=sim_lib)
    return make


def test_tiny_schedule_matches_reference_golden(factory):
    mimi_cases.check_tiny_against_golden(factory, "cpu")


@pytest.mark.parametrize("B", [1, 3])
def test_tiny_matches_oracle_with_masks_and_reset(factory, B):
    mimi_cases.oracle_vs_engine(factory, "cpu", tiny_mimi_config(), seed=21 + B, B=B, F=6, K=5)


@pytest.mark.parametrize("mtb,w,ks", [(1, 2, 2), (2, 4, 3), (4, 8, 8), (4, 1, 1)])
def test_conv_kernel_variants(factory, monkeypatch, mtb, w, ks):
    """Every tiling the planner can pick at full size (m-tiles per wave, split-K over waves / workgroups), forced onto
    the tiny codec: B=6 puts the audio-rate layers on k_conv_wide and the rest on k_pack_b_f32 + k_gemm_f32."""
    monkeypatch.setenv("MMI_CONV_MTB", str(mtb))
    monkeypatch.setenv("MMI_CONV_W", str(w))
    monkeypatch.setenv("MMI_CONV_KSPLIT", str(ks))
    mimi_cases.oracle_vs_engine(factory, "cpu", tiny_mimi_config(), seed=90 + mtb, B=6, F=3, K=5)


def test_fused_residual_block_on_the_tiny_codec(factory, monkeypatch):
    """k_resblock (both convs of a SEANet residual block in one launch, the hidden tensor handed over in the accumulator layout)
    forced onto the tiny codec's first block (96 columns per session: 32-column tiles inside one session; the other blocks' tiles
    would straddle sessions and stay on two launches) - one hidden m-tile - with exec masks, a mid-run reset and the reference's
    golden schedule."""
    monkeypatch.setenv("MMI_MIMI_RES_FUSION_MIN", "1")
    mimi_cases.check_tiny_against_golden(factory, "cpu")
    mimi_cases.oracle_vs_engine(factory, "cpu", tiny_mimi_config(), seed=31, B=6, F=4, K=5)


def test_general_shape_fallbacks_of_the_launch_mergers(factory, monkeypatch):
    """The merged launches of the benchmark shapes have general-shape fallbacks (one RVQ level per launch pair when the codec has
    no single semantic level, pack launches in front of the RVQ projections and the first transposed-conv GEMM above 128 columns,
    two commit launches when the history table does not fit by value, dense rows for short buffers); these test hooks force them
    on the tiny codec: the oracle's codes and PCM."""
    for var in ("MMI_RVQ_NO_PAIR", "MMI_MIMI_PACK_LAUNCHES", "MMI_MIMI_TWO_COMMITS", "MMI_MIMI_NO_ALIGN"):
        monkeypatch.setenv(var, "1")
    mimi_cases.oracle_vs_engine(factory, "cpu", tiny_mimi_config(), seed=41, B=3, F=4, K=5)


@pytest.mark.parametrize("heads,two_pass", [(2, False), (4, False), (2, True)])
def test_attention_head_dims_and_ring_wrap(factory, monkeypatch, heads, two_pass):
    """Head dims 64 / 32 (the tiny codec has 16) through the one-round-trip attention kernel and its two-pass fallback,
    with more steps than ring slots so that the ring wraps and old positions fall out of the context."""
    from dataclasses import replace
    if two_pass:
        monkeypatch.setenv("MMI_MIMI_ATTN_TWO_PASS", "1")
    cfg = replace(tiny_mimi_config(), dimension=128, tr_d_model=128, tr_num_heads=heads, tr_context=5)
    mimi_cases.oracle_vs_engine(factory, "cpu", cfg, seed=50 + heads, B=2, F=5, K=3)


def test_multi_frame_call_equals_frame_by_frame(factory):
    cfg = tiny_mimi_config()
    sd = random_mimi_state_dict(cfg, seed=3)
    x = 0.3 * torch.randn(2, 1, cfg.frame_size * 4, generator=torch.Generator().manual_seed(1))
    a, b = factory(sd, cfg, 4), factory(sd, cfg, 4)
    with a.streaming(2), b.streaming(2):
        whole = a.encode(x)
        parts = torch.cat([b.encode(x[..., f * cfg.frame_size:(f + 1) * cfg.frame_size]) for f in range(4)], -1)
        assert torch.equal(whole, parts)
        pw = a.decode(whole)
        pp = torch.cat([b.decode(parts[..., f:f + 1]) for f in range(4)], -1)
        assert torch.equal(pw, pp)
    # non-streaming encode == streaming from a fresh state (compression.py:354-359), incl. right padding
    c2 = a.encode(x[..., :cfg.frame_size * 3 + 5])
    assert c2.shape == (2, 4, 4) and torch.equal(c2[..., :3], whole[..., :3])


def test_error_conventions(factory):
    cfg = tiny_mimi_config()
    m = factory(random_mimi_state_dict(cfg, seed=3), cfg, 4, max_batch=2)
    with m.streaming(2):
        with pytest.raises(RuntimeError, match="multiple of the frame size"):
            m.encode(torch.zeros(2, 1, cfg.frame_size + 1))
        with pytest.raises(AssertionError):
            m.encode(torch.zeros(1, 1, cfg.frame_size))      # batch != streaming batch
        with pytest.raises(RuntimeError):
            m.streaming_forever(2)                            # already streaming (streaming.py:113)
    with pytest.raises(AssertionError):
        m.streaming_forever(3)                                # > max_batch
    sd = random_mimi_state_dict(cfg, seed=3)
    sd.pop("downsample.conv.conv.conv.weight")
    with pytest.raises(KeyError):
        factory(sd, cfg, 4)


def test_properties_mirror_reference(factory):
    cfg = tiny_mimi_config()
    m = factory(random_mimi_state_dict(cfg, seed=3), cfg, 4)
    assert (m.sample_rate, m.frame_rate, m.frame_size, m.channels) == (1200, 12.5, 96, 1)
    assert (m.num_codebooks, m.total_codebooks, m.cardinality) == (4, 5, 48)
    m.set_num_codebooks(2)
    assert m.encode(torch.zeros(1, 1, cfg.frame_size)).shape == (1, 2, 1)


def test_decode_takes_any_number_of_codebooks(sim_lib):
    """`mimi.decode` with fewer codebooks than the encoder produces (the reference's split RVQ decodes however many it is given,
    compression.py:406-429, vq.py:281-287): models whose LM generates dep_q != n_q - dep_q codebooks need it.  Against the
    oracle, switching K between calls of one stream."""
    import numpy as np
    import torch
    from moshi_amd import MimiModel, tiny_mimi_config
    from moshi_amd.weights import random_mimi_state_dict
    from oracle.mimi_oracle import MimiOracle
    cfg = tiny_mimi_config()
    sd = random_mimi_state_dict(cfg, seed=11)
    m = MimiModel(sd, cfg, device="cpu", max_batch=2, num_codebooks=4, lib=sim_lib)
    orc = MimiOracle(sd, cfg, num_codebooks=4)
    rng = np.random.default_rng(2)
    orc.streaming(2)
    with m.streaming(2):
        for K in (4, 2, 3, 4):
            codes = rng.integers(0, cfg.q_bins, (2, K, 1))
            got = m.decode(torch.from_numpy(codes)).numpy()
            want = orc.decode(codes)
            assert np.abs(got - want).max() < 1e-4, K
        with pytest.raises(AssertionError):
            m.decode(torch.zeros(2, cfg.q_n_q + 1, 1, dtype=torch.long))


def test_c2_recipe_on_the_tiny_codec(factory):
    """The C2 recipe (random exec masks, mid-run reset, rings wrapping many times over: context 6) on the simulator: the
    same case the GPU suite runs at full size for 200 frames (tests/test_a_mimi_gpu.py)."""
    res = mimi_cases.check_c2_recipe(factory, "cpu", tiny_mimi_config(), B=4, F=14, K=5)
    assert res["wrapped"]
