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


def _header_prototypes():
    """{function: (return class, [argument classes])} from the header; classes: ptr, i32, i64, u64, f32, f64, void."""
    text = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "moshi_mi.h").read_text(), flags=re.S)
    text = re.sub(r"typedef struct mmi_\w+ \{.*?\} mmi_\w+;", "", text, flags=re.S)      # struct bodies hold function pointers

    def cls(t):
        t = t.strip()
        if "*" in t or re.search(r"\bmmi_stream\b", t):
            return "ptr"
        base = re.sub(r"\b(const|unsigned(?= long)|struct)\b", "", t).split()
        base = base[0] if base else "void"
        return {"int": "i32", "int32_t": "i32", "mmi_status": "i32", "uint32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "size_t": "u64",
                "float": "f32", "double": "f64", "void": "void"}[base]
    out = {}
    def top_level_split(arglist):                                   # a function-pointer argument holds commas of its own
        parts, depth, cur = [], 0, ""
        for ch in arglist:
            depth += ch == "("
            depth -= ch == ")"
            if ch == "," and depth == 0:
                parts.append(cur)
                cur = ""
            else:
                cur += ch
        return parts + [cur]
    for m in re.finditer(r"([\w\s\*]+?)\b(mmi_[a-z0-9_]+)\s*\(([^;{}]*)\)\s*;", text):
        args = [a for a in top_level_split(m.group(3)) if a.strip() and a.strip() != "void"]
        out[m.group(2)] = (cls(m.group(1)), [cls(re.sub(r"\b\w+\s*$", "", a) if not a.strip().endswith("*") else a) for a in args])
    return out


def test_prototypes_of_the_header_equal_the_ctypes_signatures():
    """Argument count and machine class (pointer / 32-bit / 64-bit / float) of every entry point: header == binding."""
    import ctypes as C
    from moshi_amd import _capi

    def cls(t):
        if t is None:
            return "void"
        if t in (C.c_void_p, C.c_char_p) or issubclass(t, (C._Pointer, C._CFuncPtr)):
            return "ptr"
        return {C.c_int: "i32", C.c_int32: "i32", C.c_uint32: "i32", C.c_int64: "i64", C.c_uint64: "u64", C.c_size_t: "u64",
                C.c_float: "f32", C.c_double: "f64"}[t]
    protos = _header_prototypes()
    assert sorted(protos) == declared_symbols()
    bad = []
    for name, (res, args) in _capi.SIGNATURES.items():
        mine = (cls(res), [cls(a) for a in args])
        if mine != protos[name]:
            bad.append((name, "header", protos[name], "binding", mine))
    assert not bad, bad


_NULL_PROBE = r"""
import ctypes as C, sys
sys.path.insert(0, sys.argv[1])
from moshi_amd import _capi
lib = _capi.load(sys.argv[2])
for name, (res, args) in sorted(_capi.SIGNATURES.items()):
    vals = [None if (a in (C.c_void_p, C.c_char_p) or issubclass(a, (C._Pointer, C._CFuncPtr))) else (0.0 if a in (C.c_float, C.c_double) else 0)
            for a in args]
    r = getattr(lib, name)(*vals)
    print(name, r if res is C.c_int else "-", flush=True)
"""


def test_every_entry_point_survives_null_arguments(sim_lib, tmp_path):
    """"Return 0 / a negative status, never throw across the ABI" (SURVEY.md 8b): every entry point called with a NULL handle,
    NULL pointers and zero sizes returns - a status < 0 from everything that acts on a handle, 0 from the pure queries
    (`*_batch`, `*_state_bytes`, `*_launch_list`, ...) - and the process is still alive afterwards.  Host code only, so the
    simulator build of the library (the same sources) answers for the product; run in a child process so that a crash is a
    failed assertion, not a dead test worker."""
    import subprocess
    import sys
    from moshi_amd import _capi
    script = tmp_path / "null_probe.py"
    script.write_text(_NULL_PROBE)
    p = subprocess.run([sys.executable, str(script), str(ROOT), str(sim_lib.path)], capture_output=True, text=True, timeout=120)
    seen = dict(ln.split() for ln in p.stdout.splitlines() if len(ln.split()) == 2)
    assert p.returncode == 0, f"crashed after {list(seen)[-1:]}: {p.stderr[-400:]}"
    assert sorted(seen) == sorted(_capi.SIGNATURES)
    queries = {"mmi_version", "mmi_duplex_batch", "mmi_lm_has_hooks", "mmi_lm_launch_list", "mmi_lm_model_rows", "mmi_lm_profile_sites",
               "mmi_lm_state_bytes", "mmi_lm_streaming_batch", "mmi_mimi_launch_list", "mmi_mimi_num_codebooks", "mmi_mimi_state_bytes",
               "mmi_mimi_streaming_batch"}
    for name, r in seen.items():
        if r == "-":
            continue
        if name in queries:
            assert int(r) == (3 if name == "mmi_version" else 0), (name, r)
        else:
            assert int(r) < 0, f"{name} accepted a NULL handle: {r}"
