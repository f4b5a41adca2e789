"""Duplex pipeline (include/moshi_mi.h mmi_duplex_*) on the CPU kernel simulator: same bits as the serial serving loop."""
import pytest

from tests import duplex_cases


@pytest.mark.parametrize("use_sampling", [False, True])
def test_pipelined_frames_equal_the_serial_loop(sim_lib, use_sampling):
    duplex_cases.check_pipeline_is_bit_identical("cpu", sim_lib, use_sampling=use_sampling)


def test_decode_reads_a_column_slice_in_place(sim_lib):
    duplex_cases.check_strided_decode("cpu", sim_lib)


def test_duplex_refuses_models_that_are_not_streaming(sim_lib):
    from moshi_amd.duplex import DuplexStream
    from moshi_amd.lm import LMGen
    from tests.batcher_cases import tiny_pair
    mimi, lm, _, _ = tiny_pair("cpu", sim_lib, 2)
    with pytest.raises(RuntimeError):
        DuplexStream(mimi, LMGen(lm))
