"""Duplex pipeline (mmi_duplex_*) on a real MI355X: three streams + events + hipGraph replay give the same bits as the serial
serving loop, on the tiny models and on the full-size codec with an LM at Moshi-7B's widths."""
import pytest

from tests import duplex_cases

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("use_sampling", [False, True])
def test_pipelined_frames_equal_the_serial_loop(gpu_lib, use_sampling):
    duplex_cases.check_pipeline_is_bit_identical(DEV, None, B=5, steps=24, use_sampling=use_sampling, join_every=6)


def test_pipeline_crosses_the_attention_program_switch(gpu_lib, monkeypatch):
    """Fewer than 32 sessions: the LM keeps two step programs (decode attention with / without its merge launch, lm_engine.hip
    attn_variant) and moves from one to the other when its host-side bound on the ring depth passes the threshold - here after 5 of
    24 frames, with the ring split forced onto the tiny model - inside the pipeline's two-graph form of the step (run_split)."""
    monkeypatch.setenv("MMI_ATTN_NS", "3")
    monkeypatch.setenv("MMI_ATTN_SOLO", "5")
    duplex_cases.check_pipeline_is_bit_identical(DEV, None, B=5, steps=24, use_sampling=True, join_every=6)


def test_decode_reads_a_column_slice_in_place(gpu_lib):
    duplex_cases.check_strided_decode(DEV, None)


def test_full_size_codec_and_moshi_width_lm_pipelined_equal_serial(gpu_lib):
    """Real Mimi + an LM at the 7B widths (2 temporal layers), 32 sessions, 20 frames submitted without a join in between:
    tokens and PCM identical to the serial loop."""
    from moshi_amd import MimiConfig, MimiModel
    from moshi_amd.config import LMConfig
    from moshi_amd.lm import LMModel
    from moshi_amd.weights import random_lm_state_dict, random_mimi_state_dict
    mcfg, lcfg = MimiConfig(), LMConfig(num_layers=2, context=64)
    B = 32
    mimi = MimiModel(random_mimi_state_dict(mcfg, seed=3, device=DEV), mcfg, device=DEV, max_batch=B, num_codebooks=8)
    lm = LMModel(random_lm_state_dict(lcfg, seed=4, device=DEV), lcfg, device=DEV, max_batch=B)
    n = duplex_cases.check_pipeline_is_bit_identical(DEV, None, B=B, steps=20, use_sampling=True, join_every=20,
                                                     pair=(mimi, lm, mcfg, lcfg))
    assert n >= 18
