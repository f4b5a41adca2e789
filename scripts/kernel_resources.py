"""Per-kernel resource table of the gfx950 code objects in a built library: VGPRs (+ AGPRs), SGPRs, LDS bytes, scratch (spill) bytes,
the workgroup size the kernel was compiled for, the waves per SIMD its registers allow and its code size - from the AMDGPU metadata
notes and the symbol table.  No GPU needed.

    python scripts/kernel_resources.py [lib.so] > profiles/rNN_kernel_resources.csv

Needs /opt/rocm/lib/llvm/bin/{llvm-readelf} and c++filt."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from isa_scan_packed_swizzle import extract  # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin/"
FIELDS = (".agpr_count", ".vgpr_count", ".sgpr_count", ".group_segment_fixed_size", ".private_segment_fixed_size",
          ".max_flat_workgroup_size", ".vgpr_spill_count", ".sgpr_spill_count")


def kernels_of(co):
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", str(co)], capture_output=True, text=True).stdout
    sizes = {}
    for ln in subprocess.run([LLVM + "llvm-readelf", "-sW", str(co)], capture_output=True, text=True).stdout.splitlines():
        p = ln.split()
        if len(p) >= 8 and p[3] == "FUNC":
            sizes[p[7]] = int(p[2])
    out, cur = [], None
    for ln in notes.splitlines():
        if re.match(r"^  - \.", ln):            # a new entry of amdhsa.kernels
            cur = {}
            out.append(cur)
            ln = "    " + ln[4:]
        m = re.match(r"^    (\.[a-z_]+):\s+(.*)$", ln)
        if cur is not None and m and (m.group(1) in FIELDS or m.group(1) == ".name"):
            cur[m.group(1)] = m.group(2).strip().strip("'")
    out = [k for k in out if ".name" in k]
    for k in out:
        k["code_bytes"] = sizes.get(k[".name"], 0)
    return out


def waves_per_simd(vgpr, agpr):
    # gfx950: 512 unified VGPRs per SIMD lane, allocated in blocks of 8; at most 8 waves per SIMD
    total = max(1, -(-(vgpr + agpr) // 8) * 8)
    return min(8, 512 // total)


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else str(Path(__file__).resolve().parent.parent / "moshi_amd" / "libmoshi_mi.so")
    rows = []
    with tempfile.TemporaryDirectory() as d:
        for co in extract(lib, d):
            rows += kernels_of(co)
    names = subprocess.run(["c++filt"], input="\n".join(k[".name"] for k in rows), capture_output=True, text=True).stdout.splitlines()
    print("kernel,vgpr,agpr,sgpr,lds_bytes,scratch_bytes,vgpr_spills,workgroup,waves_per_simd,code_bytes")
    for k, n in sorted(zip(rows, names), key=lambda kn: kn[1]):
        v, a = int(k.get(".vgpr_count", 0)), int(k.get(".agpr_count", 0))
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(.*\)$", "", n)
        print(f"\"{n}\",{v},{a},{k.get('.sgpr_count', 0)},{k.get('.group_segment_fixed_size', 0)},{k.get('.private_segment_fixed_size', 0)},"
              f"{k.get('.vgpr_spill_count', 0)},{k.get('.max_flat_workgroup_size', 0)},{waves_per_simd(v, a)},{k['code_bytes']}")
    scr = [(n, k) for k, n in zip(rows, names) if int(k.get(".private_segment_fixed_size", 0)) > 0]
    print(f"# {len(rows)} kernels, {len(scr)} with scratch", file=sys.stderr)


if __name__ == "__main__":
    main()
