import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`: ~200 simulator / oracle / host-logic tests, 10+ minutes in one process) spreads over a few
    worker processes when pytest-xdist is there; the GPU suite never does (one GPU, timing-sensitive cases).  MMI_TEST_WORKERS=0
    (or an explicit -n) switches this off."""
    # a worker process runs this hook too (with numprocesses reset to None): it must never spawn workers of its own
    if os.environ.get("PYTEST_XDIST_WORKER") or os.environ.get("MMI_XDIST_PARENT") or hasattr(config, "workerinput"):
        return None
    want = os.environ.get("MMI_TEST_WORKERS", "")
    if want == "0" or "not gpu" not in (config.getoption("markexpr", "") or ""):
        return None
    if not config.pluginmanager.hasplugin("xdist") or getattr(config.option, "numprocesses", None):
        return None
    n = int(want) if want.isdigit() else min(6, os.cpu_count() or 1)
    if n > 1:
        os.environ["MMI_XDIST_PARENT"] = str(os.getpid())        # inherited by every worker: second guard against re-entry
        for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):    # n workers x all cores of BLAS threads each
            os.environ.setdefault(var, "2")                                            # thrash: 8.3 -> 7.1 min on 8 vCPUs
        config.option.numprocesses = n
        config.option.dist = "load"
        config.option.tx = ["popen"] * n
    return None


def pytest_collection_modifyitems(config, items):
    """CPU suite only: the one long test (the 32-layer oracle against the reference's own run: ~3 minutes of numpy on 15 GB of
    weights) starts FIRST, so that it runs beside the rest instead of after it.  The GPU suite keeps its file order (core first)."""
    if "not gpu" not in (config.getoption("markexpr", "") or ""):
        return
    items.sort(key=lambda it: 0 if "at_the_benchmark_depth" in it.nodeid else 1)      # stable: everything else keeps its place


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
    # MMI_SIM_LIB: another build of the simulator library (the AddressSanitizer build: tests/hipsim/build_sim.py build_asan)
    return _capi.load(os.environ.get("MMI_SIM_LIB") or build_sim.build())


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library on a real GPU.  Fails (never skips) if the HIP extension is missing."""
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from moshi_amd import _capi
    return _capi.load()
