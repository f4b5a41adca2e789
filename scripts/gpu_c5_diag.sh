# VERDICT r4 item 1, on one box (gpurun -- 'bash scripts/gpu_c5_diag.sh'):
#  (a) the C5 file (per-linear bit equality + the network gates), as pytest runs it;
#  (b) the 64-session int8 network case in FRESH processes, graph and eager: is the engine's output the same bits every time?
#  (c) the same case with the oracle's BLAS on 1 thread and on all: does the checker move?
#  (d) the WHOLE -m gpu suite with every handle allocation poisoned (MMI_DEBUG_POISON=1: 0xFF bytes), with per-test durations:
#      any test that changes its verdict reads state the engine never initialised.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python -m pytest tests/test_y_c5_int8_gpu.py -m gpu -q -x --durations=0 > $O/pytest_c5.log 2>&1; echo "pytest c5 rc=$?"; tail -25 $O/pytest_c5.log | cut -c1-200
cat > /tmp/c5_once.py <<'PY'
import sys, json, hashlib, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from tests import lm_cases
from moshi_amd.config import LMConfig
tag = sys.argv[1]
res = lm_cases.int8_network_vs_oracle("cuda", None, LMConfig(num_layers=2, context=64), seed=364, B=64, S=2, use_masks=True, name="c5_b64_" + tag)
print("RESULT", tag, json.dumps(res["engine_vs_oracle"]))
PY
for i in 1 2; do timeout 300 python /tmp/c5_once.py graph$i 2>&1 | grep RESULT; done
for i in 1 2; do MMI_NO_GRAPH=1 timeout 300 python /tmp/c5_once.py eager$i 2>&1 | grep RESULT; done
OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 MKL_NUM_THREADS=1 timeout 600 python /tmp/c5_once.py blas1 2>&1 | grep RESULT
MMI_DEBUG_POISON=1 timeout 300 python /tmp/c5_once.py poison 2>&1 | grep RESULT
MMI_DEBUG_POISON=1 timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --durations=0 > $O/pytest_gpu_poison.log 2>&1; echo "pytest poison rc=$?"; tail -5 $O/pytest_gpu_poison.log | cut -c1-300
grep -E "^[0-9.]+s (call|setup)" $O/pytest_gpu_poison.log | head -45
for q in none q8; do timeout 300 python bench.py --no-cpu-baseline --no-extras --batch 64 --quant $q > $O/bench_duplex_b64_$q.log 2>&1; grep '"metric"' $O/bench_duplex_b64_$q.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b64 $q ms/step %.3f' % d['ms_per_step'])"; done
timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench_default_noextras.log 2>&1; grep '"metric"' $O/bench_default_noextras.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b32 ms/step %.3f' % d['ms_per_step'])"
