"""moshi_amd - MI355X-native engine for the Mimi -> Moshi LM -> Mimi full-duplex frame step.

The arithmetic lives in `libmoshi_mi.so` (hand-written gfx950 HIP kernels behind the C ABI of
include/moshi_mi.h); this package is the thin host mirror of the reference's Python API.
"""
from .config import LMConfig, MimiConfig, tiny_lm_config, tiny_mimi_config  # noqa: F401
from .mimi import MimiModel  # noqa: F401
from .lm import ConditionFuser, LMGen, LMModel  # noqa: F401
from .batcher import SessionBatcher  # noqa: F401

__all__ = ["MimiConfig", "LMConfig", "MimiModel", "LMModel", "LMGen", "ConditionFuser", "SessionBatcher",
           "tiny_mimi_config", "tiny_lm_config"]
