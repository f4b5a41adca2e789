"""Build tests/hipsim/libmoshi_sim.so: the SAME engine/kernel sources as the product library, compiled for the
host against the hipsim fiber simulator.  TEST INFRASTRUCTURE ONLY - used by `-m "not gpu"` tests to exercise the
kernel logic (indexing, ring state, masks, packing, reductions) where no GPU exists.  The product never loads it.
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
CSRC = ROOT / "moshi_amd" / "csrc"
LIB = HERE / "libmoshi_sim.so"
SOURCES = ["api_common.hip", "mimi_engine.hip", "lm_engine.hip", "batcher.hip", "duplex.hip"]


def _cxx() -> str:
    for cand in (os.environ.get("HIPSIM_CXX"), "/opt/rocm/lib/llvm/bin/clang++", "clang++"):
        if cand and (not os.path.isabs(cand) or os.path.exists(cand)):
            return cand
    raise RuntimeError("clang++ not found")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = list(CSRC.glob("*")) + list(HERE.glob("*.h")) + list(HERE.glob("*.cpp")) + [ROOT / "include" / "moshi_mi.h"]
    return any(p.stat().st_mtime > t for p in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    """Several test workers may call this at once: the compile runs under a file lock."""
    import fcntl
    with open(HERE / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool) -> Path:
    if not force and not needs_build():
        return LIB
    objs, jobs = [], []
    common = [_cxx(), "-O2", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread", "-Wno-unused-value",
              f"-I{HERE}", f"-I{CSRC}"]
    for name in SOURCES:
        src = CSRC / name
        if not src.exists():
            continue
        obj = HERE / (src.stem + ".sim.o")
        cmd = common + ["-x", "c++", "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append((subprocess.Popen(cmd), cmd))
        objs.append(obj)
    obj = HERE / "hipsim.sim.o"
    cmd = common + ["-c", str(HERE / "hipsim.cpp"), "-o", str(obj)]
    jobs.append((subprocess.Popen(cmd), cmd))
    objs.append(obj)
    for p, cmd in jobs:
        if p.wait() != 0:
            raise RuntimeError("sim build failed: " + " ".join(cmd))
    cmd = [_cxx(), "-shared", "-fPIC", "-pthread", "-o", str(LIB)] + [str(o) for o in objs]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
