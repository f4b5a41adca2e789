"""`moshi.models` as its callers import it (moshi/moshi/models/__init__.py:9-15; server.py:24, run_inference.py:20:
`from .models import loaders, MimiModel, LMModel, LMGen`): the same names, so that switching a caller to the engine is changing
`moshi.models` to `moshi_amd.models`."""
from . import loaders  # noqa: F401
from .lm import LMGen, LMModel  # noqa: F401
from .loaders import get_mimi, get_moshi_lm  # noqa: F401
from .mimi import MimiModel  # noqa: F401

__all__ = ["MimiModel", "LMModel", "LMGen", "get_mimi", "get_moshi_lm", "loaders"]
