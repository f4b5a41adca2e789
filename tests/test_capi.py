"""The C-ABI library loads and exports every symbol include/moshi_mi.h declares (no compute without a GPU)."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "moshi_mi.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mmi_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_abi():
    syms = declared_symbols()
    for must in ("mmi_mimi_encode_step", "mmi_mimi_decode_step", "mmi_lm_step", "mmi_mimi_set_exec_mask",
                 "mmi_mimi_reset", "mmi_lm_create"):
        assert must in syms


def test_binding_covers_every_declared_symbol():
    from moshi_amd import _capi
    assert sorted(_capi.SIGNATURES) == declared_symbols()


def test_product_library_builds_loads_and_exports_all_symbols():
    """hipcc cross-compiles for gfx950 without a GPU; the library must load and export the full ABI."""
    from moshi_amd import _capi, build
    build.build(verbose=False)
    lib = _capi.load()
    for name in declared_symbols():
        assert hasattr(lib.cdll, name), name
    assert lib.mmi_version() == 3


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from moshi_amd import _capi
    monkeypatch.setattr(_capi, "DEFAULT_LIB", tmp_path / "nope.so")
    monkeypatch.setattr(_capi, "_default", None)
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _capi.load()


def test_product_model_refuses_cpu_device():
    import torch
    from moshi_amd import MimiModel, tiny_mimi_config
    with pytest.raises(RuntimeError, match="no CPU path"):
        MimiModel({}, tiny_mimi_config(), device="cpu")
