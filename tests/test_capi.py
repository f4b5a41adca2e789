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


# ---- layout level: the header's structs as gcc lays them out == the ctypes Structures of the binding ------------------------------
_STRUCTS = {"mmi_tensor_desc": "TensorDesc", "mmi_mimi_cfg": "MimiCfg", "mmi_lm_cfg": "LMCfg", "mmi_sampling": "Sampling",
            "mmi_guidance": "Guidance", "mmi_lm_hooks": "LMHooks", "mmi_batcher_cfg": "BatcherCfg",
            "mmi_batcher_stats": "BatcherStats"}


def _header_structs():
    """{struct name: [field names in order]} parsed from the header (comments stripped; array / pointer / function-pointer fields)."""
    text = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "moshi_mi.h").read_text(), flags=re.S)
    out = {}
    for m in re.finditer(r"typedef struct (mmi_\w+) \{(.*?)\} \1;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            fp = re.search(r"\(\*\s*(\w+)\)", decl)                 # function pointer: ret (*name)(args)
            if fp:
                fields.append(fp.group(1))
                continue
            for part in decl.split(","):                            # `int32_t used_slots, total_slots`
                fields.append(re.search(r"(\w+)\s*(\[\w+\])?$", part.strip()).group(1))
        out[m.group(1)] = fields
    return out


def test_header_compiles_standalone_as_c99_and_as_cxx(tmp_path):
    """`#include "moshi_mi.h"` alone must be a valid translation unit for a C caller (the boundary is plain C) and a C++ one."""
    import subprocess
    for flags, name in ((["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror"], "t.c"), (["g++", "-std=c++17", "-Wall", "-Werror"], "t.cpp")):
        src = tmp_path / name
        src.write_text('#include "moshi_mi.h"\nint main(void) { return MMI_ABI_VERSION == 3 ? 0 : 1; }\n')
        subprocess.check_call(flags + [f"-I{ROOT / 'include'}", "-fsyntax-only", str(src)])


def test_struct_layouts_of_the_header_equal_the_ctypes_binding(tmp_path):
    """The binding's Structures are written by hand: a field added to the header but not to _capi.py (or a changed type) would
    shift every later field silently.  A C program built from the header prints sizeof / offsetof of every field of every struct
    and the values of both enums; they must equal the ctypes layout and the binding's constants."""
    import ctypes
    import subprocess
    from moshi_amd import _capi
    structs = _header_structs()
    assert sorted(structs) == sorted(_STRUCTS), "a struct of the header has no ctypes Structure in the binding (or the reverse)"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "moshi_mi.h"', 'int main(void) {']
    for s, fields in structs.items():
        lines.append(f'  printf("{s} . %zu\\n", sizeof({s}));')
        for f in fields:
            lines.append(f'  printf("{s} {f} %zu\\n", offsetof({s}, {f}));')
    for e in ("MMI_OK", "MMI_ERR_INVALID", "MMI_ERR_SHAPE", "MMI_ERR_STATE", "MMI_ERR_HIP", "MMI_ERR_MISSING_WEIGHT", "MMI_ERR_UNSUPPORTED",
              "MMI_ERR_BUSY", "MMI_ERR_NO_CHANNEL", "MMI_F32", "MMI_BF16", "MMI_I64", "MMI_F16", "MMI_I8", "MMI_F8E4M3"):
        lines.append(f'  printf("enum {e} %d\\n", (int){e});')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", f"-I{ROOT / 'include'}", str(src), "-o", str(exe)])
    for ln in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines():
        s, f, v = ln.split()
        if s == "enum":
            assert getattr(_capi, f) == int(v), f"{f}: header {v}, binding {getattr(_capi, f)}"
            continue
        cls = getattr(_capi, _STRUCTS[s])
        if f == ".":
            assert ctypes.sizeof(cls) == int(v), f"sizeof({s}): header {v}, binding {ctypes.sizeof(cls)}"
            assert [n for n, *_ in cls._fields_] == structs[s], f"{s}: field names / order differ from the header"
        else:
            assert getattr(cls, f).offset == int(v), f"{s}.{f}: offset {v} in the header, {getattr(cls, f).offset} in the binding"
