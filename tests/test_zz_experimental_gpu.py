"""GPU tests of code paths that are OFF by default (enabled by environment switches), kept in the file pytest collects last so
that `pytest -x` has run every default-path parity test before it gets here."""
import pytest

from moshi_amd.config import LMConfig
from tests import lm_cases

pytestmark = pytest.mark.gpu
DEV = "cuda"


# mode "2" (each tile's epilogue under the last chunk's weight stream) has only run on the simulator so far
# (tests/test_lm_sim.py); it joins the default list once it has been timed and checked on hardware:
# MMI_TEST_XLDS_MODES=1,2 python -m pytest tests/test_zz_experimental_gpu.py -m gpu   (scripts/gpu_next_xlds.sh)
import os

_MODES = os.environ.get("MMI_TEST_XLDS_MODES", "1").split(",")


@pytest.mark.parametrize("B,mode", [(B, m) for m in _MODES for B in (18, 40)])
def test_lds_resident_gemm_full_width_matches_oracle(gpu_lib, monkeypatch, B, mode):
    """MMI_GEMM_LDS=1: the temporal in_proj / gated linear_in and the grouped depformer_in run on k_gemm_xlds (activations
    staged in LDS, one workgroup per CU walking 1-3 n-tiles) at the 7B layer shapes, one and two batch tiles, vs the oracle;
    mode "2": each tile's epilogue under the last chunk's weight stream."""
    monkeypatch.setenv("MMI_GEMM_LDS", mode)
    st = {}
    lm_cases.oracle_vs_engine(DEV, None, LMConfig(num_layers=2, context=64), seed=15, B=B, S=2, use_masks=False, stats=st)
    assert st["xlds_launches"] >= 2 * 2 + 1


