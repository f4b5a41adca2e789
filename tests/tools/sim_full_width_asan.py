"""One-off: production-shape cases on the AddressSanitizer build of the simulator (every hipMalloc its own heap block): a kernel
reading or writing past a buffer at the REAL sizes is reported with its source line.

    python tests/hipsim/build_sim.py --asan /tmp/asan
    LD_PRELOAD=<libclang_rt.asan-x86_64.so> ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
    MMI_SIM_LIB=/tmp/asan/libmoshi_sim_asan.so python tests/tools/sim_full_width_asan.py        (~2 h on 8 vCPUs)"""
import os, sys, time
os.environ["MMI_NO_GRAPH"] = "1"
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests' / 'hipsim'))
from moshi_amd import _capi, MimiModel, MimiConfig
from moshi_amd.config import LMConfig
lib = _capi.load(os.environ["MMI_SIM_LIB"])
from tests import lm_cases, mimi_cases
def run(name, fn):
    t = time.time()
    fn()
    print(f"== {name}: ok ({time.time() - t:.0f} s)", flush=True)
cfg = LMConfig(num_layers=2, context=64)
def factory(sd, c, K, max_batch=8):
    return MimiModel(sd, c, device="cpu", max_batch=max_batch, num_codebooks=K, lib=lib)
run("full-size Mimi against the reference golden (2 sessions, 2 frames)", lambda: mimi_cases.check_full_against_golden(factory, "cpu"))
run("7B layer shapes against the reference golden (16-row tiles)", lambda: lm_cases.check_golden_wide("cpu", lib))
run("18 sessions at full width against the oracle (k_gemm_xlds, 32-row tile)", lambda: lm_cases.oracle_vs_engine("cpu", lib, cfg, seed=15, B=18, S=2, use_masks=False))
run("40 sessions int8 x int8 at full width (two batch tiles)", lambda: lm_cases.oracle_vs_engine("cpu", lib, cfg, seed=340, B=40, S=2, use_masks=True, quantize=True))
run("ring wrap at the real capacity (3000 slots, seek to 2995 / 2990, 14 steps)", lambda: lm_cases.ring_wrap_at_real_capacity("cpu", lib, B=2, S=14))
