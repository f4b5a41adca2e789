"""Scan the gfx950 code objects of a built library for packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32)
that CROSS-swizzle their halves (`op_sel:[..]`) while writing one of their own source pairs in place - the instruction form that
stood in k_gemm_xp's RoPE epilogue when its output was not reproducible.

    python scripts/isa_scan_packed_swizzle.py [lib.so]

Prints the kernels that contain one.  Needs /opt/rocm/lib/llvm/bin/llvm-objdump."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from device_code_hash import code_objects  # noqa: E402,F401  (same bundle walk)
import struct  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
PAT = re.compile(r"\s(v_pk_(?:mul|add|fma)_f32)\s+v\[(\d+):(\d+)\],\s*([^/]*)")


def extract(path, out_dir):
    data = Path(path).read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    files = []
    for m in re.finditer(magic, data):
        i = m.start()
        p = i + len(magic)
        num = struct.unpack_from("<Q", data, p)[0]
        p += 8
        for _ in range(num):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tlen].decode()
            p += tlen
            if "gfx950" in triple and size > 0:
                f = Path(out_dir) / f"co_{len(files)}.co"
                f.write_bytes(data[i + off:i + off + size])
                files.append(f)
    return files


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else str(Path(__file__).resolve().parent.parent / "moshi_amd" / "libmoshi_mi.so")
    hits = {}
    with tempfile.TemporaryDirectory() as d:
        for co in extract(lib, d):
            asm = subprocess.run([OBJDUMP, "-d", str(co)], capture_output=True, text=True).stdout
            cur = None
            for ln in asm.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:", ln)
                if m:
                    cur = m.group(1)
                    continue
                m = PAT.search(ln)
                if not m or "op_sel:[" not in m.group(4):
                    continue
                if any(a == m.group(2) for a, _ in re.findall(r"v\[(\d+):(\d+)\]", m.group(4))):
                    hits.setdefault(cur, []).append(ln.split("//")[0].strip())
    for k, v in hits.items():
        print(f"{len(v):3d}  {k}")
        for x in v[:3]:
            print("       " + x)
    print(f"{len(hits)} kernel(s) with an in-place cross-swizzled packed-fp32 instruction in {lib}")


if __name__ == "__main__":
    main()
