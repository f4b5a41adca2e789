# Round 4, GPU call B: the DPP attention against the chunked kernel, and C5 on the int8 matrix core (int8 activations).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
line() { python - "$1" <<'PY'
import sys, json
try:
    d = json.loads([l for l in open(sys.argv[1]) if '"metric"' in l][-1])
    s = d.get('roofline', {}).get('sites', {})
    def g(k): return s.get(k, {}).get('us_per_op', float('nan'))
    print('ms/step %.3f p50 %.3f frames/s %.0f step-frac %.3f | norm1 %.1f in_proj %.1f attn %.1f out_proj %.1f norm2 %.1f ffn_in %.1f ffn_out %.1f | dom %.1f us %.3f' % (
        d['ms_per_step'], d.get('p50_ms_per_step', 0), d['value'], d.get('roofline', {}).get('step', {}).get('frac', 0),
        g('L.norm1'), g('L.in_proj'), g('L.attn'), g('L.out_proj'), g('L.norm2'), g('L.ffn_in'), g('L.ffn_out'),
        1e3 * d.get('roofline', {}).get('avg_launch_ms', 0), d.get('roofline', {}).get('frac', 0)))
except Exception as e:
    print('no line:', e)
PY
}
rm -f $O/b_summary.txt
timeout 900 python -m pytest tests/test_lm_gpu.py -m gpu -q -x --timeout=600 -s -k "int8" > $O/pytest_int8.log 2>&1; echo "pytest int8 rc=$?" | tee -a $O/b_summary.txt; grep -a "passed\|failed\|Error\|assert" $O/pytest_int8.log | tail -6
for attn in split wave; do
  MMI_ATTN=$attn timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 8 > $O/b_mid_${attn}.log 2>&1
  echo "duplex depth=mid attn=$attn: $(line $O/b_mid_${attn}.log)" | tee -a $O/b_summary.txt
  MMI_ATTN=$attn timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --kv-depth full --steps 20 --warmup 5 > $O/b_full_${attn}.log 2>&1
  echo "lm full context attn=$attn: $(line $O/b_full_${attn}.log)" | tee -a $O/b_summary.txt
done
timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --batch 1 --steps 40 --warmup 8 > $O/b_lm_b1.log 2>&1; echo "lm B=1: $(line $O/b_lm_b1.log)" | tee -a $O/b_summary.txt
# C5: 64 sessions, int8 linears: weight-only (rounds 1-3) / int8 activations on k_gemm_xp / on k_gemm_xlds; fp8 and bf16 beside them
MMI_Q8_ACT=bf16 timeout 240 python bench.py --no-cpu-baseline --no-extras --batch 64 --quant q8 --steps 40 --warmup 8 > $O/b_b64_q8_wonly.log 2>&1; echo "b64 q8 weight-only: $(line $O/b_b64_q8_wonly.log)" | tee -a $O/b_summary.txt
timeout 240 python bench.py --no-cpu-baseline --no-extras --batch 64 --quant q8 --steps 40 --warmup 8 > $O/b_b64_q8_act8.log 2>&1; echo "b64 q8 int8 activations: $(line $O/b_b64_q8_act8.log)" | tee -a $O/b_summary.txt
MMI_GEMM_LDS=1 timeout 240 python bench.py --no-cpu-baseline --no-extras --batch 64 --quant q8 --steps 40 --warmup 8 > $O/b_b64_q8_act8_xlds.log 2>&1; echo "b64 q8 int8 activations, k_gemm_xlds: $(line $O/b_b64_q8_act8_xlds.log)" | tee -a $O/b_summary.txt
timeout 240 python bench.py --no-cpu-baseline --no-extras --batch 64 --steps 40 --warmup 8 > $O/b_b64_bf16.log 2>&1; echo "b64 bf16: $(line $O/b_b64_bf16.log)" | tee -a $O/b_summary.txt
timeout 240 python bench.py --no-cpu-baseline --no-extras --batch 32 --quant q8 --steps 40 --warmup 8 > $O/b_b32_q8_act8.log 2>&1; echo "b32 q8 int8 activations: $(line $O/b_b32_q8_act8.log)" | tee -a $O/b_summary.txt
