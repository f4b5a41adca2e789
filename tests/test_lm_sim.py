"""LMGen.step kernels on the CPU kernel simulator against the reference golden vectors and the numpy oracle."""
import numpy as np
import pytest
import torch

from moshi_amd.config import tiny_lm_config
from moshi_amd.weights import random_lm_state_dict
from tests import lm_cases


def test_greedy_schedule_matches_reference_golden(sim_lib):
    lm_cases.check_golden_greedy("cpu", sim_lib)


def test_free_running_greedy_follows_the_reference_golden(sim_lib):
    res = lm_cases.check_golden_tiny_free_running("cpu", sim_lib)
    assert res["token_decisions_compared"] > 50


def test_sampled_run_matches_reference_golden_given_its_noise(sim_lib):
    lm_cases.check_golden_sampled("cpu", sim_lib)


def test_in_kernel_sampler_follows_the_reference_rule(sim_lib):
    lm_cases.engine_sampling_matches_oracle_rule("cpu", sim_lib)


def test_sampling_without_top_k_is_a_multinomial_over_the_whole_vocabulary(sim_lib):
    lm_cases.check_full_multinomial("cpu", sim_lib)


@pytest.mark.parametrize("top_k,top_k_text,card", [(20, 10, None), (1, 3, None), (60, 90, None), (250, 25, (2048, 8200))])
def test_production_sampler_token_for_token(sim_lib, top_k, top_k_text, card):
    """top-k + the engine's own counter RNG (no supplied noise): every sampled token recomputed from the logits taps - the
    threshold found through the window of 8 high bytes under the largest key, ties at the threshold, one Philox call per four
    vocabulary entries; k = 1; a set that reaches into the negative logits, i.e. below the window (60 of 64: the generic two-pass
    select takes over); the default k (250 / 25) on vocabularies of the real size classes."""
    cfg = tiny_lm_config()
    if card:
        from dataclasses import replace
        cfg = replace(cfg, card=card[0], text_card=card[1])
    lm_cases.check_topk_device_rng("cpu", sim_lib, cfg, top_k=top_k, top_k_text=top_k_text, steps=3, B=2 if card else 3)


def test_sampler_large_vocabulary_variants(sim_lib):
    """Vocabularies above 2048 / 8192 entries take the wider / uncached sampler kernels (lm_kernels.h k_sample)."""
    from dataclasses import replace
    cfg = replace(tiny_lm_config(), text_card=8200, card=2056)
    lm_cases.engine_sampling_matches_oracle_rule("cpu", sim_lib, cfg, top_k=30, top_k_text=25, B=2, steps=2)


@pytest.mark.parametrize("B", [1, 3])
def test_matches_oracle_with_masks_and_reset(sim_lib, B):
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=40 + B, B=B, S=5)


@pytest.mark.parametrize("B", [18, 34])
def test_wide_batch_tiles_match_oracle(sim_lib, B):
    """17..32 sessions use the 32x32x16 MFMA tile, 33..64 two batch tiles per weight fragment (lm_kernels.h)."""
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=60 + B, B=B, S=3)


@pytest.mark.parametrize("B", [18, 32])
def test_depth_transformer_on_its_own_tile(sim_lib, monkeypatch, B):
    """MMI_DEP_TILE=16 at 17..32 sessions with bf16 weights: the depth transformer on a 16-row tile (mmi_lm::Td; two batch tiles
    per 16-row weight tile, k_gemm_xp<16, 2, ..> / k_gemm_xp_norm<16, 2, ..>) while the temporal transformer keeps the 32-row
    tile.  An opt-in (measured slower inside the step, lm_engine.hip mmi_lm::Td); checked against the oracle at the edges of the
    range so that the per-weight tile (GemmW::T) stays honest."""
    st = {}
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=160 + B, B=B, S=3, stats=st)
    assert st["dep_tile"] == 32
    monkeypatch.setenv("MMI_DEP_TILE", "16")
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=160 + B, B=B, S=3, stats=st)
    assert st["dep_tile"] == 16


@pytest.mark.parametrize("B,tile", [(18, None), (18, "16"), (3, None)])
def test_gemm_with_the_whole_slice_requested_at_once(sim_lib, monkeypatch, B, tile):
    """k_gemm_xp_once (a wave's whole K-slice in flight before its first MFMA: the depth transformer's linear_out at the 7B
    widths) forced onto every bf16 GEMM of the tiny model whose slices fit (MMI_GEMM_ONCE=a): residual, embedding, split-K
    partial, RoPE / ring and row-major epilogues, octet sharing, 16- and 32-row tiles, one and two batch tiles."""
    monkeypatch.setenv("MMI_GEMM_ONCE", "a")
    monkeypatch.setenv("MMI_GEMM_LDS", "0")
    if tile:
        monkeypatch.setenv("MMI_DEP_TILE", tile)
    st = {}
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=170 + B, B=B, S=3, stats=st)
    assert sum("k_gemm_xp_once" in k for _, k in st["launch_list"]) >= 20
    monkeypatch.setenv("MMI_GEMM_KSPLIT", "2")
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=171 + B, B=B, S=2, stats=st)


@pytest.mark.parametrize("kernel", ["wave", "wave-solo", "wave-switch", "wave-kernel", "split"])
def test_attention_ring_split_over_several_workgroups(sim_lib, monkeypatch, kernel):
    """Fewer (session, head) pairs than CUs (one real-time session: 32 pairs): the ring is shared out over several workgroups
    per pair and their partial (max, sum, output) merged by k_lm_attn_combine - forced here on the tiny model, for the
    barrier-free kernel of round 4 ("wave") and the chunked one it replaces ("split").  The wave kernel's other paths
    (lm_engine.hip attn_variant): "wave-solo" = rings of up to MMI_ATTN_SOLO rows, workgroup 0 alone and no merge at all;
    "wave-switch" = the engine changes from that step program to the one with the merge launch when its host-side bound on the
    ring depth passes the threshold (here after 4 of the 9 steps); "wave-kernel" = the merge done by the last workgroup to
    arrive, the shallow program's safety net."""
    monkeypatch.setenv("MMI_ATTN_NS", "3")
    monkeypatch.setenv("MMI_ATTN", kernel.split("-")[0])
    monkeypatch.setenv("MMI_ATTN_SOLO", {"wave-solo": "768", "wave-switch": "4"}.get(kernel, "0"))
    if kernel == "wave-kernel":
        monkeypatch.setenv("MMI_ATTN_MERGE", "kernel")
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=61, B=2, S=9)


def test_depformer_attention_inside_out_proj_for_one_session(sim_lib, monkeypatch):
    """One session, bf16 weights: the depth transformer's attention of micro-steps >= 1 runs inside its out_proj
    (k_dep_attn_out_proj; no dep.attn site in the launch list); MMI_NO_DEP_ATTN_FUSION=1 and more sessions keep the launch."""
    st = {}
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=73, B=1, S=5, stats=st)
    assert "dep.attn" not in st["launch_sites"]
    monkeypatch.setenv("MMI_NO_DEP_ATTN_FUSION", "1")
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=73, B=1, S=5, stats=st)
    assert st["launch_sites"]["dep.attn"] == 7 * 2
    monkeypatch.delenv("MMI_NO_DEP_ATTN_FUSION")
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=77, B=2, S=2, stats=st)
    assert st["launch_sites"]["dep.attn"] == 7 * 2


def test_split_k_gemm_with_fused_residual_norm(sim_lib, monkeypatch):
    """The K-split GEMM path (fp32 partials folded into the residual stream by k_resid_rmsnorm) that the 4096-wide
    layers take at 17..64 sessions, forced onto the tiny shapes."""
    monkeypatch.setenv("MMI_GEMM_KSPLIT", "2")
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=77, B=18, S=3)
    monkeypatch.setenv("MMI_GEMM_KSPLIT", "4")
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=78, B=3, S=3)


@pytest.mark.parametrize("mode", ["0", "2a", "4a"])
def test_small_gemms_shared_in_row_octets(sim_lib, monkeypatch, mode):
    """GemmArgs::osplit: an n-tile shared by 2 / 4 workgroups, each loading and writing only its row octets (the depth
    transformer's N = 1024 linears, 32 tiles on 256 CUs).  The default (no variable) already shares every GEMM of <= 64 tiles;
    here: off, and 2 / 4 parts forced on every eligible GEMM, at the 32-row tile (18 sessions), with two batch tiles (34) and at
    the 16-row tile (3 sessions: two octets per tile)."""
    monkeypatch.setenv("MMI_GEMM_OSPLIT", mode)
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=91, B=18, S=2)
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=92, B=34, S=2)
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=93, B=3, S=2)


def test_depformer_attention_launch_at_micro_step_0_is_optional(sim_lib, monkeypatch):
    """By default the depth transformer's first micro-step has no attention launch (softmax over one position: the output is v,
    written by in_proj's epilogue together with the frame cache); MMI_DEP_ATTN0_LAUNCH=1 keeps the launch.  Both against the
    oracle, with one and two batch tiles and at the 16-row tile; and both forms give the same greedy tokens."""
    import numpy as np
    import torch
    from moshi_amd.lm import LMGen, LMModel
    from moshi_amd.weights import random_lm_state_dict
    cfg = tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=5)

    def tokens():
        gen = LMGen(LMModel(sd, cfg, device="cpu", max_batch=3, lib=sim_lib), use_sampling=False, support_out_of_sync=True)
        rng = np.random.default_rng(0)
        with gen.streaming(3):
            return np.stack([gen.step(torch.from_numpy(rng.integers(0, cfg.card, (3, 8, 1)))).numpy() for _ in range(4)])
    fused = tokens()
    monkeypatch.setenv("MMI_DEP_ATTN0_LAUNCH", "1")
    assert np.array_equal(tokens(), fused)
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=94, B=18, S=2)
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=95, B=3, S=2)


@pytest.mark.parametrize("B,grid,mode", [(18, 8, "1"), (34, 8, "1"), (20, 64, "1"), (18, 8, "2"), (34, 8, "2"), (18, 5, "2"), (18, 7, "1")])
def test_gemm_with_activations_resident_in_lds(sim_lib, monkeypatch, B, grid, mode):
    """k_gemm_xlds (one workgroup walking several n-tiles with the activation chunks staged in LDS) on the tiny
    shapes: gated FFN input, in_proj with RoPE / ring write and the row-major heads, one and two batch tiles, 1..3 tiles per
    workgroup (grid 8: 22 gated tiles -> 2-3 whole tiles per workgroup, 12 in_proj tiles -> 6 row octets = 1.5 tiles each, the
    shared tile written half by each owner; grids 5 / 7: uneven octet shares, ranges that would span 4 tiles fall back to
    k_gemm_xp; grid 64 > tiles: one tile each)."""
    monkeypatch.setenv("MMI_GEMM_LDS", mode)     # "2": each tile's epilogue under the last chunk's weight stream
    monkeypatch.setenv("MMI_GEMM_LDS_GRID", str(grid))
    st = {}
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=140 + B, B=B, S=3, stats=st)
    # per step: 2 temporal layers x (in_proj + gated linear_in) + the text head + 8 audio heads take the kernel
    assert st["xlds_launches"] >= 3 * (2 * 2 + 1)


@pytest.mark.parametrize("B,quantize", [(18, True), (34, True), (18, "fp8"), (34, "fp8")])
def test_gemm_with_activations_resident_in_lds_quantised_weights(sim_lib, monkeypatch, B, quantize):
    """k_gemm_xlds on int8 / fp8 weight entries (two k-steps per entry, two activation fragments from LDS per entry); the tiny
    shapes give chunks of 4 entries: only 4 of the 8 waves own k-steps, the others just join the barriers and the epilogue."""
    monkeypatch.setenv("MMI_GEMM_LDS", "1")
    monkeypatch.setenv("MMI_GEMM_LDS_GRID", "8")
    st = {}
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=150 + B, B=B, S=2, quantize=quantize, stats=st)
    assert st["xlds_launches"] >= 2 * (2 * 2 + 1)


def test_depformer_in_per_step_launches(sim_lib, monkeypatch):
    """The engine normally runs the dep_q `depformer_in` linears as one grouped GEMM and lets each sampler add its token's
    embedding row; depth widths that are not whole n-tiles fall back to one GEMM per micro-step with the embedding in its
    epilogue.  Force that path."""
    monkeypatch.setenv("MMI_NO_DEP_IN_GROUP", "1")
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=91, B=3, S=4)
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=92, B=2, S=3, quantize=True)


@pytest.mark.parametrize("B", [2, 18, 34])
def test_int8_linears_on_the_int8_matrix_core_match_the_bitsandbytes_restatement(sim_lib, B):
    """`quantize=True` (BASELINE configs[4]): the reference's QLinear.forward is bitsandbytes' int8 x int8 matmul
    (utils/quantize.py:24-40) - activations quantised row-wise (absmax / 127, round half even), int32 accumulation on
    v_mfma_i32_*_i8, dequantised by SCA * SCB / 127^2.  Against the oracle's restatement of that published rule (bnb itself is
    absent: unpinned against the library), same tolerance as the bf16 path; 2 sessions = the 16-row tile (16x16x64), 18 / 34
    = one / two batch tiles of the 32-row tile (32x32x32)."""
    st = {}
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=80 + B, B=B, S=3, quantize=True, stats=st)
    assert st["launch_sites"].get("L.out_proj", 0) >= 2      # (quantisation launch + GEMM per layer)


@pytest.mark.parametrize("B", [2, 18])
def test_int8_weight_only_mode_matches_its_oracle(sim_lib, monkeypatch, B):
    """MMI_Q8_ACT=bf16: the weight-only form of rounds 1-3 (int8 weights widened to bf16 in registers, bf16 activations),
    against the oracle in the same mode."""
    monkeypatch.setenv("MMI_Q8_ACT", "bf16")
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=80 + B, B=B, S=3, quantize=True, int8_activations=False)


def test_int8_activations_without_norm_fusion_and_with_split_k(sim_lib, monkeypatch):
    """The un-fused norm (its own launch stores the int8 row) and the K-split temporal GEMMs on int8 operands."""
    monkeypatch.setenv("MMI_NO_NORM_FUSION", "1")
    monkeypatch.setenv("MMI_GEMM_KSPLIT", "2")
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=87, B=18, S=3, quantize=True)


def test_int8_activations_on_the_lds_resident_gemm(sim_lib, monkeypatch):
    monkeypatch.setenv("MMI_GEMM_LDS", "1")
    monkeypatch.setenv("MMI_GEMM_LDS_GRID", "8")
    st = {}
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=88, B=34, S=2, quantize=True, stats=st)
    assert st["xlds_launches"] >= 2 * (2 * 2 + 1)


@pytest.mark.parametrize("B", [3, 18, 40])
def test_int8_linear_bit_exact_per_linear_against_the_oracle(sim_lib, B):
    """`QLinear.forward` of ONE module (utils/quantize.py:24-40) through the step's kernels
    (`mmi_lm_debug_linear`) - int8 codes, row absmax and bf16 output IDENTICAL to the oracle's restatement, for every linear
    family of the model (k_quant_rows_i8 + k_gemm_xp, the norm launch's int8 copy, k_gemm_q8 with and without the norm), at the
    16-row tile, one and two batch tiles of the 32-row tile."""
    checked = lm_cases.int8_linears_bit_exact("cpu", sim_lib, tiny_lm_config(), B, seed=500 + B)
    assert len(checked) >= 20


@pytest.mark.parametrize("B", [18, 40])
def test_int8_linear_bit_exact_on_the_split_k_and_lds_resident_forms(sim_lib, monkeypatch, B):
    """The same per-linear bit equality with the full-size plans forced onto the tiny shapes: the K-split GEMM whose int32
    partials the norm launch adds and dequantises (out_proj / linear_out at the 7B widths), and k_gemm_xlds on int8 operands."""
    monkeypatch.setenv("MMI_GEMM_KSPLIT", "2")
    monkeypatch.setenv("MMI_GEMM_LDS", "1")
    monkeypatch.setenv("MMI_GEMM_LDS_GRID", "8")
    checked = lm_cases.int8_linears_bit_exact("cpu", sim_lib, tiny_lm_config(), B, seed=520 + B)
    assert any("[splitk]" in n for n, _ in checked)


def test_bitsandbytes_rule_restated_known_answers():
    """The oracle's restatement of bitsandbytes' int8 rule on hand-computed cases: absmax scaling, round HALF TO EVEN, an
    all-zero row, exact integer accumulation beyond 2^24, the dequantisation constant."""
    import numpy as np
    from oracle.lm_oracle import QWeight, int8_vectorwise_quant, linear_int8
    x = np.array([[127.0, 63.5, -0.5, 1.5, 2.5, -2.5], [0, 0, 0, 0, 0, 0], [1.0, -2.0, 0.25, 0.5, -0.5, 0.75]], np.float32)
    ca, sca = int8_vectorwise_quant(x)
    assert sca.ravel().tolist() == [127.0, 0.0, 2.0]
    assert ca[0].tolist() == [127, 64, 0, 2, 2, -2]                 # 63.5 -> 64 (even), -0.5 -> -0, 1.5 -> 2, 2.5 -> 2, -2.5 -> -2
    assert ca[1].tolist() == [0] * 6
    assert ca[2].tolist() == [64, -127, 16, 32, -32, 48]            # x * 63.5: 63.5 -> 64, 15.875 -> 16, 31.75 -> 32, 47.625 -> 48
    K = 4096
    q = np.full((2, K), 127, np.float32); q[1] = -127
    w = QWeight(q, np.array([2.0, 4.0], np.float32))
    w.act8 = True
    y = linear_int8(np.full((1, K), 3.0, np.float32), w)            # CA = 127 everywhere: out32 = +-127 * 127 * 4096 = 66 064 384 > 2^24
    exact = np.array([[K * 3.0 * 2.0, -K * 3.0 * 4.0]])
    assert np.allclose(y, exact, rtol=4e-3) and abs(float(y[0, 0]) - exact[0, 0]) <= 2 ** -8 * exact[0, 0]      # only the bf16 rounding of y


@pytest.mark.parametrize("B,input_scale", [(2, 1.0), (18, 0.25), (34, 1.0)])
def test_fp8_weights_on_the_fp8_mfma_match_the_fp8_oracle(sim_lib, B, input_scale):
    """`quantize="fp8"` (BASELINE configs[4]): e4m3fn linears with per-row scales, activations converted to e4m3 in registers,
    v_mfma_*_fp8_fp8.  Same tolerance as the bf16 path against an oracle that holds the same fp8 tensors and quantises the
    activations at the same points."""
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=90 + B, B=B, S=3, quantize="fp8", input_scale=input_scale)


@pytest.mark.parametrize("B,S,quantize", [(2, 15, False), (18, 3, False), (3, 4, "fp8")])
def test_fp8_kv_ring_matches_the_oracle(sim_lib, B, S, quantize):
    """`kv_cache_dtype="fp8"` (SURVEY.md 8d C5 "fp8 KV"): keys (after RoPE) and values enter the ring as e4m3 bytes, the decode
    attention widens them exactly; S > context: the ring wraps.  Oracle: the same e4m3 rounding at the ring write."""
    from dataclasses import replace
    lm_cases.oracle_vs_engine("cpu", sim_lib, replace(tiny_lm_config(), kv_cache_dtype="fp8"), seed=120 + B, B=B, S=S, quantize=quantize)


def test_fp8_hardware_yardstick_runs_on_the_simulator(sim_lib):
    """The check the GPU tests hold the fp8 engine to (tests/lm_cases.py "fp8 on hardware"): on the simulator, whose MFMA
    accumulates exactly, the engine coincides with the exact oracle, far inside the yardstick."""
    msg = lm_cases.fp8_engine_within_format_conditioning("cpu", sim_lib, tiny_lm_config(), seed=92, B=2, S=2)
    assert msg.startswith("engine: max-rel 0.000/0.0")


def test_fp8_without_norm_fusion_and_with_split_k(sim_lib, monkeypatch):
    monkeypatch.setenv("MMI_NO_NORM_FUSION", "1")
    monkeypatch.setenv("MMI_GEMM_KSPLIT", "2")
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=97, B=18, S=3, quantize="fp8")


def test_e4m3_rounding_of_oracle_matches_torch_float8():
    """The oracle's e4m3 rounding (what the engine's v_cvt_pk_fp8_f32 does on gfx950: OCP e4m3fn, round-to-nearest-even,
    saturating) against torch's own float8_e4m3fn cast on every bf16 value in range and on exact ties."""
    from oracle.lm_oracle import e4m3r
    bits = np.arange(0, 1 << 16, dtype=np.uint32)
    x = (bits << 16).astype(np.uint32).view(np.float32)
    x = x[np.isfinite(x) & (np.abs(x) <= 448.0)]
    ties = np.array([0.0009765625, 0.0029296875, 0.017578125, 18.0, 22.0, 416.0, 448.0, 464.0, 1e4, -1e4], np.float32)
    for v in (x, ties):
        ref = torch.from_numpy(np.clip(v, -448, 448)).to(torch.float8_e4m3fn).float().numpy()
        assert np.array_equal(e4m3r(v), ref)


def test_fp8_quantisation_error_and_storage():
    from moshi_amd.weights import quantize_lm_state_dict_fp8
    cfg = tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=4)
    q = quantize_lm_state_dict_fp8(sd, input_scale=0.5)
    k = "transformer.layers.0.gating.linear_in.weight"
    assert q[k].dtype == torch.float8_e4m3fn and q[k + "_scale"].shape == (sd[k].shape[0],)
    assert q["transformer.layers.0.gating.linear_in.input_scale"].item() == 0.5
    assert q["emb.0.weight"].dtype == torch.bfloat16
    deq = q[k].float() * q[k + "_scale"][:, None]
    w = sd[k].float()
    assert ((deq - w).abs() <= w.abs() / 16 + q[k + "_scale"][:, None] * 2.0 ** -9).all()   # 3 mantissa bits: half a step <= 1/16


def test_int8_quantisation_error_is_small_and_storage_is_bnb_style():
    from moshi_amd.weights import quantize_lm_state_dict
    cfg = tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=4)
    q = quantize_lm_state_dict(sd)
    k = "transformer.layers.0.gating.linear_in.weight"
    assert q[k].dtype == torch.int8 and q[k + "_scb"].dtype == torch.float32 and q[k + "_scb"].shape == (sd[k].shape[0],)
    assert q["emb.0.weight"].dtype == torch.bfloat16 and "emb.0.weight_scb" not in q          # embeddings are not quantised
    deq = q[k].float() * (q[k + "_scb"] / 127.0)[:, None]
    w = sd[k].float()
    assert (deq - w).abs().max() <= 0.51 * (q[k + "_scb"] / 127.0).max() + 1e-3 * w.abs().max()   # half a step (+ the fp16 cast)
    assert int(q[k].abs().max()) == 127


def test_depformer_replace_tokens_argument(sim_lib):
    """`LMGen.step(codes, depformer_replace_tokens=...)` (lm.py:751-755): the given audio tokens enter the delay ring instead
    of the depformer's own; checked against the oracle teacher-forced on the same audio tokens."""
    from oracle.lm_oracle import LMOracle
    cfg = tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=33)
    B = 2
    gen = lm_cases.make_engine(cfg, sd, "cpu", sim_lib, B, use_sampling=False, support_out_of_sync=True)
    orc = LMOracle(sd, cfg)
    orc.streaming(B)
    rng = np.random.default_rng(7)
    with gen.streaming(B):
        for s in range(5):
            codes = rng.integers(0, cfg.card, (B, 8, 1))
            repl = rng.integers(0, cfg.card, (B, cfg.dep_q, 1))
            forced = np.concatenate([np.full((B, 1), -1), repl[:, :, 0]], 1)
            oo, _ = orc.step(codes, use_sampling=False, forced=forced, support_out_of_sync=True)
            out = gen.step(torch.from_numpy(codes), depformer_replace_tokens=torch.from_numpy(repl))
            assert np.array_equal(out.numpy()[:, 1:], oo[:, 1:]), f"step {s}: audio rows of the ring differ"


def test_none_during_delay_and_errors(sim_lib):
    g = np.load(lm_cases.GOLDEN / "lm_tiny.npz")
    cfg = tiny_lm_config()
    gen = lm_cases.make_engine(cfg, random_lm_state_dict(cfg, seed=17), "cpu", sim_lib, 3, use_sampling=False)
    with pytest.raises(RuntimeError, match="streaming"):
        gen.step(torch.zeros(3, 8, 1, dtype=torch.long))
    with gen.streaming(3):
        pattern = [gen.step(torch.from_numpy(g["codes"][s])) is None for s in range(3)]
        assert pattern == list(g["none_pattern"])            # lm.py:774-776
        with pytest.raises(AssertionError):
            gen.step(torch.zeros(2, 8, 1, dtype=torch.long))  # batch mismatch (lm.py:681)
        with pytest.raises(AssertionError):
            gen.step(torch.zeros(3, 7, 1, dtype=torch.long))  # too few user codebooks (lm.py:683-686)
        out = gen.step(torch.zeros(3, 9, 1, dtype=torch.long))  # extra rows are ignored (lm.py:688-689)
        assert out.shape == (3, 9, 1) and out.dtype == torch.int64
    with pytest.raises(AssertionError, match="no fuser"):   # lm.py:600-603
        lm_cases.LMGen(gen.lm_model, cfg_coef=2.0)
    # check=True (lm.py:703-711): an ungenerated (-2) or out-of-range user code is an AssertionError, the zero token (-1) is not
    chk = lm_cases.LMGen(gen.lm_model, use_sampling=False, check=True)
    with chk.streaming(3):
        ok = torch.from_numpy(g["codes"][0]).clone()
        ok[0, 0, 0] = -1
        chk.step(ok)
        for bad in (-2, cfg.card + 1):
            codes = ok.clone()
            codes[1, 2, 0] = bad
            with pytest.raises(AssertionError):
                chk.step(codes)


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_guidance_conditioning_and_extra_heads_match_reference_golden(sim_lib, name):
    """classifier-free guidance (a: masked-until, b: no-text, c: condition tensors), a `sum`
    condition and extra heads (d), against the reference's own LMGen runs (tests/golden/lm_cfg.npz)."""
    lm_cases.check_cfg_engine("cpu", sim_lib, name)


def test_asr_style_model_without_depformer_matches_reference_golden(sim_lib):
    """dep_q = 0 (the reference's stt models: lm.py:218-221): the step ends at the text sampler, tokens are [B, 1, 1], the
    extra heads read the transformer output; against the reference's own run (tests/golden/lm_stt.npz)."""
    lm_cases.check_stt_engine("cpu", sim_lib)


def test_guided_batch_must_fit_twice(sim_lib):
    cfg = tiny_lm_config()
    lm = lm_cases.LMModel(random_lm_state_dict(cfg, seed=1), cfg, device="cpu", max_batch=3, lib=sim_lib)
    gen = lm_cases.LMGen(lm, cfg_coef=2.0, cfg_is_no_text=True)
    with pytest.raises(AssertionError, match="two model rows per session"):
        gen.streaming_forever(2)


def test_get_and_set_streaming_state_resume_a_dialogue(sim_lib):
    lm_cases.check_streaming_state_snapshot("cpu", sim_lib)


def test_seek_moves_sessions_and_the_ring_wraps(sim_lib):
    """mmi_lm_seek (test / benchmark aid): sessions jump to positions just before the ring capacity of an all-zero ring and
    step across the wrap; engine and oracle agree.  Also the launch list of the step names a site for every kernel."""
    from oracle.lm_oracle import LMOracle
    cfg = tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=12)
    gen = lm_cases.make_engine(cfg, sd, "cpu", sim_lib, 2, use_sampling=False, support_out_of_sync=True)
    orc = LMOracle(sd, cfg)
    orc.streaming(2)
    start = np.array([cfg.context - 2, 3 * cfg.context - 1])
    rng = np.random.default_rng(3)
    with gen.streaming(2):
        assert gen.launch_list() == []
        gen.seek(start); orc.seek(start)
        for s in range(4):
            codes = rng.integers(0, cfg.card, (2, 8, 1))
            oo, (otl, oal, ott, oat) = orc.step(codes, use_sampling=False, support_out_of_sync=True)
            forced = np.concatenate([ott[:, None], oat], 1)
            out, tl, al = gen.step_with_taps(torch.from_numpy(codes), forced_tokens=torch.from_numpy(forced))
            assert np.array_equal(out.numpy(), oo)
            for b in range(2):
                assert lm_cases.logits_close(tl[b].numpy(), otl[b])
        ll = gen.launch_list()
    sites = [s for s, _ in ll]
    assert sites[0] == "prepare" and sites[-1] == "commit" and "-" not in sites
    assert sites.count("L.ffn_in") == cfg.num_layers and sites.count("dep.out_proj") == cfg.dep_q * cfg.depformer_num_layers
    assert all(k.startswith("k_") for _, k in ll)


def test_full_depth_checker_plumbing_on_the_tiny_model(sim_lib):
    """The machinery of the 32-layer GPU parity test (lazy fp32 widening of the temporal linears, rows at different depths,
    per-site error log) on the tiny model, where it runs in seconds."""
    from dataclasses import replace
    log = lm_cases.full_depth_vs_oracle("cpu", sim_lib, B=3, S=2, seed=9, name="plumbing", cfg=replace(tiny_lm_config(), context=100))
    assert set(log.summary()) == {"text"} | {f"audio{k}" for k in range(tiny_lm_config().dep_q)}


def test_rng_sampling_statistics(sim_lib):
    """The sampler's production form (on-device RNG, no rank computation) on the simulator: a coarser frequency check."""
    lm_cases.rng_sampling_statistics("cpu", sim_lib, iters=3, tol=0.13)


def test_step_hooks_see_and_modify_the_step_like_the_reference(sim_lib):
    lm_cases.check_step_hooks("cpu", sim_lib)


def test_a_raising_hook_surfaces_its_exception(sim_lib):
    cfg = tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=22)
    lm = lm_cases.LMModel(sd, cfg, device="cpu", max_batch=1, lib=sim_lib)

    def bad(_):
        raise ValueError("boom")
    gen = lm_cases.LMGen(lm, use_sampling=False, on_text_hook=bad)
    with gen.streaming(1):
        with pytest.raises(ValueError, match="boom"):
            gen.step(torch.zeros(1, 8, 1, dtype=torch.long))


@pytest.mark.parametrize("name", ["e", "f"])
def test_cross_attention_conditioning_matches_reference_golden(sim_lib, name):
    """A model with cross-attention layers (transformer.py:727-732, 779-797) fed by ConditionFuser.get_cross (base.py:392-409):
    `e` cross + sum conditions, `f` guidance with two concatenated cross tensors and the sinusoidal position embedding."""
    lm_cases.check_cross_engine("cpu", sim_lib, name)


def test_cross_attention_more_positions_than_slots(sim_lib):
    """T_c larger than the 16 position slots of one pass (head dim 32), 18 sessions (32-row tiles): the online softmax across
    passes and the batched key / value projection at stream start."""
    from dataclasses import replace
    lm_cases.cross_vs_oracle("cpu", sim_lib, replace(tiny_lm_config(), cross_attention=True), B=18, S=2, Tc=41, seed=8)


def test_int8_depformer_batch_tiles_walked_by_one_workgroup(sim_lib, monkeypatch):
    """MMI_Q8_TILES=serial: k_gemm_q8<32, 2, ..> (both batch tiles through one workgroup's registers, the round-4 form kept for
    same-box A/Bs) stays bit-identical to the oracle; the default at 33..64 sessions is one batch tile per workgroup (grid.y)."""
    monkeypatch.setenv("MMI_Q8_TILES", "serial")
    lm_cases.int8_linears_bit_exact("cpu", sim_lib, tiny_lm_config(), 40, seed=561)
    lm_cases.oracle_vs_engine("cpu", sim_lib, tiny_lm_config(), seed=562, B=40, S=2, quantize=True)


def test_debug_trace_of_two_sessions_is_identical(sim_lib, monkeypatch, tmp_path):
    """MMI_DEBUG_TRACE (the tool that found round 5's unreproducible launch): every op followed by a checksum of every
    allocation of the streaming state.  Two sessions of one handle that start from identical memory (MMI_DEBUG_POISON=0) and are
    fed the same frames write the same file - on the simulator by construction; on MI355X it is the property under test."""
    monkeypatch.setenv("MMI_DEBUG_TRACE", str(tmp_path / "trace"))
    monkeypatch.setenv("MMI_DEBUG_POISON", "0")
    monkeypatch.setenv("MMI_DEBUG_DUMP", "1:2:10")           # q of the temporal in_proj, step 1
    from moshi_amd.weights import quantize_lm_state_dict
    cfg = tiny_lm_config()
    sd = quantize_lm_state_dict(random_lm_state_dict(cfg, seed=3))
    gen = lm_cases.make_engine(cfg, sd, "cpu", sim_lib, 5, use_sampling=False, support_out_of_sync=True)
    rng = np.random.default_rng(0)
    codes = [torch.from_numpy(rng.integers(0, cfg.card, (5, 8, 1))) for _ in range(2)]
    for _ in range(2):
        with gen.streaming(5):
            for c in codes:
                gen.step(c)
    files = sorted(tmp_path.glob("trace.*"))
    traces = [f for f in files if f.suffix[1:].isdigit()]
    assert len(traces) == 2 and traces[0].read_text() == traces[1].read_text() and len(traces[0].read_text().splitlines()) > 100
    dumps = sorted(tmp_path.glob("trace.*.a10"))
    assert len(dumps) == 2 and dumps[0].read_bytes() == dumps[1].read_bytes() and any(dumps[0].read_bytes())
