import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def sim_lib():
    """The kernel simulator build of the engine (tests/hipsim) - CPU-side check of the kernel logic."""
    os.environ["MMI_NO_GRAPH"] = "1"   # hipsim has no graph capture; the launch list runs eagerly
    sys.path.insert(0, str(ROOT / "tests" / "hipsim"))
    import build_sim
    from moshi_amd import _capi
    return _capi.load(build_sim.build())


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library on a real GPU.  Fails (never skips) if the HIP extension is missing."""
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from moshi_amd import _capi
    return _capi.load()
