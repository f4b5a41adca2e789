"""GPU tests of the GEMM paths that are NOT the default at the 7B layer shapes (selected by environment switches), kept in the
file pytest collects last so that `pytest -x` has run every default-path parity test before it gets here."""
import pytest

from moshi_amd.config import LMConfig
from tests import lm_cases

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("B,mode", [(18, "0"), (40, "0"), (40, "1")])      # (one batch tile on the plain-tail k_gemm_xlds: the simulator's)
def test_non_default_gemm_paths_at_full_width_match_oracle(gpu_lib, monkeypatch, B, mode):
    """MMI_GEMM_LDS=0: the temporal in_proj / gated linear_in and the grouped depformer_in on k_gemm_xp (one workgroup per
    n-tile, activations re-read from L2) - the fallback of the default k_gemm_xlds; MMI_GEMM_LDS=1: k_gemm_xlds with the
    plain (un-staggered) tails.  7B layer shapes, one and two batch tiles, against the oracle."""
    monkeypatch.setenv("MMI_GEMM_LDS", mode)
    st = {}
    lm_cases.oracle_vs_engine(DEV, None, LMConfig(num_layers=2, context=64), seed=15, B=B, S=2, use_masks=False, stats=st)
    assert (st["xlds_launches"] >= 2 * 2 + 1) if mode == "1" else st["xlds_launches"] == 0


def test_gemm_xp_with_rope_epilogue_is_bit_reproducible_at_two_batch_tiles(gpu_lib, monkeypatch):
    """MMI_GEMM_LDS=0 at 64 sessions, bf16: the temporal in_proj on k_gemm_xp<32, 2, ..> with the RoPE + ring-write epilogue - the
    launch whose q / k output changed from run to run before round 5 (packed-math rotation, lm_kernels.h) - repeated streams equal
    the first bit for bit."""
    monkeypatch.setenv("MMI_GEMM_LDS", "0")
    lm_cases.reproducible_between_streams(DEV, None, LMConfig(num_layers=2, context=64), B=64, quantize=False, seed=15)
