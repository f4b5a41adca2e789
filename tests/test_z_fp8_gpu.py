"""BASELINE configs[4] on a real MI355X, part 2: e4m3 linears on the fp8 MFMA and the fp8 KV ring (engine options; the reference
has no fp8 path, so the gates are the format's own conditioning - tests/lm_cases.py "fp8 on hardware").  Collected after every
core-path file and after the int8 file: `pytest -x` reaches it last but for the experimental GEMM plans."""
import numpy as np
import pytest
import torch

from moshi_amd.config import LMConfig, tiny_lm_config
from tests import lm_cases

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("B,S", [(2, 15), (18, 4), (40, 3)])
def test_fp8_kv_ring_matches_the_oracle(gpu_lib, B, S):
    """`kv_cache_dtype="fp8"`: e4m3 keys / values in the ring (written by in_proj's epilogue, widened exactly by the decode
    attention), bf16 weights, all three batch tilings; S > context: the ring wraps."""
    from dataclasses import replace
    lm_cases.oracle_vs_engine(DEV, None, replace(tiny_lm_config(), kv_cache_dtype="fp8"), seed=120 + B, B=B, S=S)


def test_fp8_hardware_primitives_match_their_definition(gpu_lib, tmp_path):
    """scripts/fp8_probe.hip on this GPU: v_cvt_pk_fp8_f32 bit-exact against the software e4m3 rounding the oracle uses
    (every bf16 value + ties), subnormal inputs honoured by the fp8 MFMA, and its dot product within 5e-4 of exact (it is
    NOT exact: small products are aligned to the group's largest - the figure tests/lm_cases.py FP8_HW_ACC_NOISE is taken from)."""
    import re
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "fp8_probe"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-w", f"-I{root / 'moshi_amd' / 'csrc'}",
                           str(root / "scripts" / "fp8_probe.hip"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120, check=True).stdout
    assert re.search(r"cvt: \d+ values, 0 mismatches", out), out
    for shape in ("32x32x16", "16x16x32"):
        err = float(re.search(rf"mfma {shape} fp8: worst \|err\| / sum\|products\| = ([0-9.e+-]+)", out).group(1))
        assert err < 5e-4, out
    got, want = map(float, re.search(r"subnormal A x 1.0: got ([0-9.e+-]+) expected ([0-9.e+-]+)", out).groups())
    assert got == want, out


@pytest.mark.parametrize("B,input_scale", [(2, 1.0), (18, 0.25), (40, 1.0)])
def test_fp8_engine_is_within_the_conditioning_of_the_fp8_network(gpu_lib, B, input_scale):
    """BASELINE configs[4]'s fp8 MFMA GEMMs (`quantize="fp8"`) on the tiny model, all three batch tilings: ring outputs exact;
    logits as close to the exact-accumulation fp8 oracle as that oracle stays to itself under the hardware's measured
    accumulate error, and as accurate against the bf16 model as the fp8 oracle (tests/lm_cases.py, "fp8 on hardware")."""
    print(lm_cases.fp8_engine_within_format_conditioning(DEV, None, tiny_lm_config(), seed=90 + B, B=B, S=3, input_scale=input_scale))


def test_fp8_full_width_within_the_conditioning_of_the_fp8_network(gpu_lib):
    """The same at the 7B layer shapes (2 temporal layers, full depformer), B=3."""
    print(lm_cases.fp8_engine_within_format_conditioning(DEV, None, LMConfig(num_layers=2, context=64), seed=10, B=3, S=2))


def test_c5_shape_fp8_linears_at_64_sessions_within_the_conditioning_of_the_fp8_network(gpu_lib):
    """BASELINE configs[4]'s own shape: 64 sessions, 7B layer widths (2 temporal layers), e4m3 linears on the fp8 MFMA."""
    print(lm_cases.fp8_engine_within_format_conditioning(DEV, None, LMConfig(num_layers=2, context=64), seed=464, B=64, S=2))


@pytest.mark.parametrize("B", [18, 40])
def test_fp8_step_is_bit_reproducible_between_streams(gpu_lib, B):
    """fp8 linears on the fp8 MFMA (one and two batch tiles), 7B layer widths: the hardware's dot-product unit is not an exact fp32
    accumulation, but it is a FUNCTION of its inputs - repeated streams on one handle equal the first bit for bit."""
    lm_cases.reproducible_between_streams(DEV, None, LMConfig(num_layers=2, context=64), B=B, quantize="fp8", seed=41 + B, repeats=3)
