"""Session batcher on the CPU kernel simulator: slot allocation, per-row reset, stream masks, routing."""
from tests import batcher_cases


def test_batcher_matches_the_schedule_driven_by_hand(sim_lib):
    batcher_cases.check_batcher_matches_manual_api("cpu", sim_lib)


def test_session_is_independent_of_its_neighbours(sim_lib):
    batcher_cases.check_session_independent_of_neighbours("cpu", sim_lib)


def test_slots_and_buffer_limits(sim_lib):
    batcher_cases.check_slots_and_buffers("cpu", sim_lib)


def test_guided_sessions_through_the_batcher(sim_lib):
    batcher_cases.check_batcher_with_guidance("cpu", sim_lib)
