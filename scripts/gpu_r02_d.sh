# round 2, fourth GPU call: 64 sessions (BASELINE configs[4] batch) - the LDS-resident GEMM against k_gemm_xp for bf16, int8 and
# fp8 linears (same box), hardware parity of the quantised LDS-resident forms at the 7B layer shapes
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cat > /tmp/xlds_q.py <<'PY'
import sys
sys.path.insert(0, '.')
import os
from moshi_amd.config import LMConfig
from tests import lm_cases
for q in (True, "fp8"):
    for B in (18, 40):
        st = {}
        os.environ["MMI_GEMM_LDS"] = "1"
        try:
            if q == "fp8":
                lm_cases.fp8_engine_within_format_conditioning("cuda", None, LMConfig(num_layers=2, context=64), seed=15, B=B, S=2)
            else:
                lm_cases.oracle_vs_engine("cuda", None, LMConfig(num_layers=2, context=64), seed=15, B=B, S=2, use_masks=False, quantize=q, stats=st)
            print("xlds quantised parity", q, B, "ok", st)
        except Exception as e:
            print("xlds quantised parity", q, B, "FAILED", repr(e)[:300])
PY
timeout 600 python /tmp/xlds_q.py > $O/xlds_quantised_parity.log 2>&1
for cfg in "none 0" "none 2" "q8 0" "q8 1" "fp8 0" "fp8 1" "none 0" "none 2" "q8 0" "q8 1"; do
  set -- $cfg
  MMI_GEMM_LDS=$2 timeout 300 python bench.py --batch 64 --quant $1 --no-cpu-baseline > $O/b64_$1_lds$2.log 2>&1
  echo "B=64 quant=$1 MMI_GEMM_LDS=$2 $(grep '"metric"' $O/b64_$1_lds$2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f frames/s %.0f dominant %.2f us' % (d['ms_per_step'], d['p50_ms_per_step'], d['value'], 1e3*d['roofline']['avg_launch_ms']))")" >> $O/b64_sweep.txt
done
cat $O/xlds_quantised_parity.log | tail -8; cat $O/b64_sweep.txt
