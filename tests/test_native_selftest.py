"""scripts/native_selftest.cpp - the Python-free self-test of an installed libmoshi_mi.so (tiny Mimi + tiny LM through the C ABI,
weights generated in the program, outputs compared with the oracle's recorded ones) - built against the CPU simulator and run:
that checks the program, the weight generator shared with tests/golden/make_native_selftest.py, and the recorded expectations
where there is no GPU.  On an MI355X the same source builds with hipcc against the product library
(profiles/r05_logs/native_selftest_mi355x.txt: the round's final library on a fresh box, PASSED)."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HIPSIM = ROOT / "tests" / "hipsim"
FIXTURE = ROOT / "tests" / "golden" / "native_selftest"


def test_native_selftest_passes_on_the_simulator_build(sim_lib, tmp_path):
    sys.path.insert(0, str(HIPSIM))
    import build_sim
    exe = tmp_path / "native_selftest_sim"
    lib = Path(sim_lib.path)
    subprocess.check_call([build_sim._cxx(), "-O1", "-w", "-std=c++17", "-ffp-contract=off", "-pthread", "-DMMI_SELFTEST_SIM", f"-I{HIPSIM}",
                           f"-I{ROOT / 'include'}", str(ROOT / "scripts" / "native_selftest.cpp"), str(lib), f"-Wl,-rpath,{lib.parent}",
                           "-o", str(exe)])
    p = subprocess.run([str(exe), str(FIXTURE)], capture_output=True, text=True, timeout=300, env={"MMI_NO_GRAPH": "1"})
    assert p.returncode == 0 and "SELFTEST PASSED" in p.stdout, p.stdout + p.stderr
    assert "0 of 30 code indices differ" in p.stdout and "0 of 90 token-ring outputs differ" in p.stdout
    # 18 sessions: the 32-row batch tile of the LM and the wide-batch conv kernels of the codec (native_selftest_b18/)
    p = subprocess.run([str(exe), str(FIXTURE.parent / "native_selftest_b18")], capture_output=True, text=True, timeout=300, env={"MMI_NO_GRAPH": "1"})
    assert p.returncode == 0 and "SELFTEST PASSED" in p.stdout, p.stdout + p.stderr
    assert "0 of 270 code indices differ" in p.stdout and "0 of 810 token-ring outputs differ" in p.stdout


def test_recorded_expectations_are_what_the_oracle_computes_today(tmp_path, monkeypatch):
    """The committed manifest regenerates byte for byte and expected.bin value for value (the oracle or the weight rule did not
    drift under them)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_native_selftest", ROOT / "tests" / "golden" / "make_native_selftest.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(mod, "OUT", tmp_path)
    mod.main()
    import numpy as np
    manifest = (FIXTURE / "manifest.txt").read_text()
    assert (tmp_path / "manifest.txt").read_text() == manifest                    # configs, every tensor's rule and seed, the layout
    new, old = (tmp_path / "expected.bin").read_bytes(), (FIXTURE / "expected.bin").read_bytes()
    assert len(new) == len(old)
    for ln in manifest.splitlines():
        if not ln.startswith("E "):
            continue
        _, name, dt, count, off = ln.split()
        dtype = np.int64 if dt == "i64" else np.float32
        a = np.frombuffer(new, dtype, int(count), int(off))
        b = np.frombuffer(old, dtype, int(count), int(off))
        if dt == "i64":
            assert np.array_equal(a, b), name                                      # codes and tokens: exact
        else:                                                                      # numpy's BLAS may order a sum differently with another
            assert np.allclose(a, b, rtol=0, atol=1e-3 * np.abs(b).max()), name    # thread count: far inside the self-test's tolerances
