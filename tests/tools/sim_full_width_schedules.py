"""One-off (too slow for the suite): the BENCHMARK's kernels at the 7B layer widths on the simulator, under perturbed wave / lane
schedules - repeated streams on one handle must stay bit-identical (tokens, logits, hidden states) when the schedule changes.

    python tests/tools/sim_full_width_schedules.py        (~7 min on 8 vCPUs)"""
import os, sys, time
os.environ["MMI_NO_GRAPH"] = "1"
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests' / 'hipsim'))
import build_sim
from moshi_amd import _capi
from moshi_amd.config import LMConfig
lib = _capi.load(build_sim.build())
from tests import lm_cases
SCHED = [(1, 1), (2, 7)]
def perturb(r):
    lib.cdll.hipsim_set_schedule(*SCHED[r])
    print(f"   repeat {r}: schedule {SCHED[r]}", flush=True)
cfg = LMConfig(num_layers=2, context=64)
for name, kw in (("32 sessions bf16 (k_gemm_xlds, split-K, 32-row tile: the benchmark's kernels)", dict(B=32, quantize=False, seed=15)),
                 ("40 sessions int8 x int8 (two batch tiles: k_gemm_xp<32, 2> with the RoPE epilogue, k_gemm_q8 per tile)", dict(B=40, quantize=True, seed=16))):
    t = time.time()
    lib.cdll.hipsim_set_schedule(0, 1)
    print(f"== {name}", flush=True)
    lm_cases.reproducible_between_streams("cpu", lib, cfg, steps=2, repeats=len(SCHED), before_repeat=perturb, **kw)
    print(f"   bit-identical under every schedule ({time.time() - t:.0f} s)", flush=True)
lib.cdll.hipsim_set_schedule(0, 1)
t = time.time()
lm_cases.oracle_vs_engine("cpu", lib, cfg, seed=15, B=18, S=2, use_masks=False)
print(f"== 18 sessions against the oracle at full width on the simulator: ok ({time.time() - t:.0f} s)", flush=True)
