"""BASELINE configs[4] on a real MI355X, part 1: row-wise int8 linears run the reference's way - `QLinear.forward` =
bitsandbytes' int8 x int8 matmul (utils/quantize.py:24-40) on v_mfma_i32_*_i8.  Collected AFTER every core-path file
(test_a_mimi, test_b_lm, test_c_*): `pytest -x` has run the codec and the bf16 LM before it gets here.

The oracle for this arithmetic is a restatement of bitsandbytes' published rule (the library is absent from /root/reference and
from this image): PARITY UNPINNED AGAINST bitsandbytes ITSELF.  What is pinned: per linear, the engine's int8 codes, row absmax
and bf16 output are IDENTICAL to that restatement; at network level a statistical gate with a yardstick (tests/lm_cases.py)."""
import numpy as np
import pytest
import torch

from moshi_amd.config import LMConfig, tiny_lm_config
from tests import lm_cases

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("B", [3, 18, 40])
def test_int8_linear_bit_exact_per_linear_tiny(gpu_lib, B):
    """Every linear family of the model through `mmi_lm_debug_linear` at the 16-row tile, one and two batch tiles of the 32-row
    tile: int8 codes == oracle codes, absmax equal, bf16 output equal bit for bit (gated linears: up to the device's expf)."""
    assert len(lm_cases.int8_linears_bit_exact(DEV, None, tiny_lm_config(), B, seed=500 + B)) >= 20


@pytest.mark.parametrize("B", [3, 18, 64])
def test_int8_linear_bit_exact_per_linear_at_the_7b_shapes(gpu_lib, B):
    """The GEMM shapes C5 runs - 4096 -> 12288 (in_proj), 4096 -> 2 x 11264 gated, 11264 -> 4096 and 4096 -> 4096 split over K with
    the int32 partials folded by the norm launch, 4096 -> 32000, the depth transformer's 1024 -> 3072 / 1024 -> 1024 / 1024 -> 2 x
    2816 / 2816 -> 1024 / 1024 -> 2048 through k_gemm_q8 with and without the norm - on k_gemm_xp<.., WQ = 3> and k_gemm_q8, 16- and
    32-row tiles, one and two batch tiles: bit for bit against the oracle (reference utils/quantize.py:24-40)."""
    report = []
    lm_cases.int8_linears_bit_exact(DEV, None, LMConfig(num_layers=1, context=64), B, seed=540 + B, report=report, weights_seed=540)
    assert any("[norm_fused]" in n for n, _ in report) and any("[fused]" in n for n, _ in report)
    assert B <= 16 or any("[splitk]" in n for n, _ in report)        # the 16-row tile (<= 16 sessions) does not split these GEMMs over K


def test_int8_linear_bit_exact_on_the_lds_resident_gemm_at_the_7b_shapes(gpu_lib, monkeypatch):
    """k_gemm_xlds<.., WQ = 3> (MMI_GEMM_LDS=1: int8 operand chunks resident in LDS) on the wide temporal GEMMs, two batch tiles."""
    monkeypatch.setenv("MMI_GEMM_LDS", "1")
    lm_cases.int8_linears_bit_exact(DEV, None, LMConfig(num_layers=1, context=64), 40, seed=580, weights_seed=540)


@pytest.mark.parametrize("B", [2, 18, 40])
def test_int8_weights_match_the_int8_oracle(gpu_lib, B):
    """C5's weight format (`quantize=True`: row-wise int8 + `weight_scb`) run the reference's way - int8 activations on
    v_mfma_i32_{16x16x64,32x32x32}_i8, bitsandbytes' row-wise rule (utils/quantize.py:24-40, restated in oracle/lm_oracle.py;
    unpinned against the library itself) - on the tiny model, all three batch tilings."""
    lm_cases.oracle_vs_engine(DEV, None, tiny_lm_config(), seed=80 + B, B=B, S=3, quantize=True)


def test_int8_weight_only_mode_matches_its_oracle(gpu_lib, monkeypatch):
    """MMI_Q8_ACT=bf16: the weight-only form of rounds 1-3 stays selectable (same-box A/Bs)."""
    monkeypatch.setenv("MMI_Q8_ACT", "bf16")
    lm_cases.oracle_vs_engine(DEV, None, tiny_lm_config(), seed=98, B=18, S=3, quantize=True, int8_activations=False)


def test_int8_full_width_layers_match_oracle(gpu_lib):
    """int8 linears at the 7B layer shapes (2 temporal layers, full depformer), B=3 with masks."""
    cfg = LMConfig(num_layers=2, context=64)
    lm_cases.oracle_vs_engine(DEV, None, cfg, seed=9, B=3, S=3, use_masks=True, quantize=True)


def test_c5_shape_int8_linears_at_64_sessions_match_oracle(gpu_lib):
    """BASELINE configs[4]'s own shape: 64 sessions, 7B layer widths (2 temporal layers), row-wise int8 linears."""
    res = lm_cases.int8_network_vs_oracle(DEV, None, LMConfig(num_layers=2, context=64), seed=364, B=64, S=2, use_masks=True, name="c5_int8_b64")
    print(res)


def test_c5_int8_engine_against_the_reference_golden_at_the_7b_layer_shapes(gpu_lib):
    """SURVEY.md 8c's definition of C5 parity, on a PINNED anchor: the int8 x int8 engine (weights quantised from the golden's
    seed) against the reference's own bf16 logits at the 7B layer shapes (tests/golden/lm_wide.npz), within a stated quantisation
    tolerance, and no further from them than 1.5 x the int8 oracle is (lm_cases.int8_engine_vs_bf16_reference; the 32-layer
    golden: tests/test_b_lm_gpu.py, where the benchmark model's weights are already drawn)."""
    lm_cases.int8_engine_vs_bf16_reference(DEV, None, "wide")
    lm_cases.int8_engine_vs_bf16_reference(DEV, None, "wide", max_batch=40, name="c5_int8_vs_reference_wide_two_batch_tiles")


def test_c5_int8_network_at_full_depth_and_64_sessions_matches_oracle(gpu_lib):
    """The benchmark model itself under C5: 32 temporal layers, 64 sessions, int8 x int8, two steps with masks and a partial reset:
    re-quantisation noise compounds per layer, and the statistical gate (engine no further from the int8 oracle than the oracle
    with fp64 statistics is) holds at full depth as it does on two layers."""
    res = lm_cases.int8_network_vs_oracle(DEV, None, LMConfig(context=64), seed=3264, B=64, S=2, use_masks=True, name="c5_int8_b64_32_layers",
                                          on_device_draw=True)
    print(res)


def test_c5_step_is_bit_reproducible_between_streams(gpu_lib):
    """64 sessions, int8 x int8, 7B layer widths: four more streams on the same handle, fed the same frames, reproduce the first
    one's tokens, logits and hidden states bit for bit.  (Round 4's driver failure - row 17 - was a launch that did not.)"""
    lm_cases.reproducible_between_streams(DEV, None, LMConfig(num_layers=2, context=64), B=64, quantize=True, seed=364, repeats=12)
