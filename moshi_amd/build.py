"""Build libmoshi_mi.so (HIP, gfx950 only) in-tree with hipcc.

    python -m moshi_amd.build            # build if sources are newer than the library
    python -m moshi_amd.build --force

The library has no torch dependency: it is a plain C-ABI shared object (include/moshi_mi.h) that
PyTorch-ROCm tensors are handed to by raw device pointer.
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
ROOT = PKG.parent
LIB = PKG / "libmoshi_mi.so"
SOURCES = ["api_common.hip", "mimi_engine.hip", "lm_engine.hip", "batcher.hip", "duplex.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def sources() -> list[Path]:
    return [CSRC / s for s in SOURCES if (CSRC / s).exists()]


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = list(CSRC.glob("*")) + [ROOT / "include" / "moshi_mi.h"]
    return any(p.stat().st_mtime > t for p in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    """Several ranks of one node may call this at once (one process per GPU): the compile runs under a file lock, and a rank
    that waited for it finds the library up to date."""
    import fcntl
    with open(PKG / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool) -> Path:
    if not force and not needs_build():
        return LIB
    objs = []
    jobs = []
    for src in sources():
        obj = CSRC / (src.stem + ".o")
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-Wno-unused-result", "-Wno-unused-value", f"-I{CSRC}", "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append((subprocess.Popen(cmd), cmd))
        objs.append(obj)
    for p, cmd in jobs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB)] + [str(o) for o in objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
