# Round 4, GPU call C: k_dep_layer (persistent depth-transformer layers) - bit-identity at the 7B widths, then the same-box A/B;
# the int8-activation path again after the absmax atomics became test-then-max.
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
    dep = sum(v.get('us_per_step', 0) for k, v in s.items() if k.startswith('dep.') or k == 'text_sample')
    print('ms/step %.3f p50 %.3f frames/s %.0f step-frac %.3f | in_proj %.1f attn %.1f out_proj %.1f ffn_in %.1f ffn_out %.1f | dep phase (live) %.0f us, dep.layer %.1f' % (
        d['ms_per_step'], d.get('p50_ms_per_step', 0), d['value'], d.get('roofline', {}).get('step', {}).get('frac', 0),
        g('L.in_proj'), g('L.attn'), g('L.out_proj'), g('L.ffn_in'), g('L.ffn_out'), dep, g('dep.layer')))
except Exception as e:
    print('no line:', e)
PY
}
rm -f $O/c_summary.txt
timeout 600 python -m pytest tests/test_lm_gpu.py -m gpu -q -x --timeout=500 -s -k "persistent_depth" > $O/pytest_persist.log 2>&1; echo "pytest persistent rc=$?" | tee -a $O/c_summary.txt; grep -a "passed\|failed\|Error\|assert" $O/pytest_persist.log | tail -6
for p in 0 1 0 1; do
  MMI_DEP_PERSIST=$p timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 8 > $O/c_duplex_p$p.log 2>&1
  echo "duplex B=32 mid persist=$p: $(line $O/c_duplex_p$p.log)" | tee -a $O/c_summary.txt
done
for p in 0 1; do
  MMI_DEP_PERSIST=$p timeout 200 python bench.py --no-cpu-baseline --no-extras --serial --steps 40 --warmup 8 > $O/c_serial_p$p.log 2>&1
  echo "serial B=32 mid persist=$p: $(line $O/c_serial_p$p.log)" | tee -a $O/c_summary.txt
  MMI_DEP_PERSIST=$p timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --batch 1 --steps 40 --warmup 8 > $O/c_lm_b1_p$p.log 2>&1
  echo "lm B=1 persist=$p: $(line $O/c_lm_b1_p$p.log)" | tee -a $O/c_summary.txt
done
timeout 300 python -m pytest tests/test_lm_gpu.py -m gpu -q -x --timeout=280 -k "int8" > $O/pytest_int8_c.log 2>&1; echo "pytest int8 rc=$?" | tee -a $O/c_summary.txt
timeout 240 python bench.py --no-cpu-baseline --no-extras --batch 64 --quant q8 --steps 40 --warmup 8 > $O/c_b64_q8_act8.log 2>&1; echo "b64 q8 int8 activations: $(line $O/c_b64_q8_act8.log)" | tee -a $O/c_summary.txt
MMI_Q8_ACT=bf16 timeout 240 python bench.py --no-cpu-baseline --no-extras --batch 64 --quant q8 --steps 40 --warmup 8 > $O/c_b64_q8_wonly.log 2>&1; echo "b64 q8 weight-only: $(line $O/c_b64_q8_wonly.log)" | tee -a $O/c_summary.txt
