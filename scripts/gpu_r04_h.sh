# Round 4, GPU call H: A/B of the interleaved K/V ring (MMI_KV_INTERLEAVE=1) at 32 and 64 sessions, mid-run depth and full context.
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
    print('ms/step %.3f p50 %.3f frames/s %.0f step-frac %.3f | in_proj %.1f attn %.1f out_proj %.1f ffn_in %.1f ffn_out %.1f' % (
        d['ms_per_step'], d.get('p50_ms_per_step', 0), d['value'], d.get('roofline', {}).get('step', {}).get('frac', 0),
        g('L.in_proj'), g('L.attn'), g('L.out_proj'), g('L.ffn_in'), g('L.ffn_out')))
except Exception as e:
    print('no line:', e)
PY
}
rm -f $O/h_summary.txt
MMI_KV_INTERLEAVE=1 timeout 400 python -m pytest tests/test_lm_gpu.py -m gpu -q -x --timeout=380 -k "ring_wrap or tiny_matches or fp8_kv or full_width_layers" > $O/pytest_h.log 2>&1; echo "pytest (interleaved ring) rc=$?" | tee -a $O/h_summary.txt; tail -1 $O/pytest_h.log
for i in 0 1 0 1; do
  MMI_KV_INTERLEAVE=$i timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 8 > $O/h_b32_i$i.log 2>&1; echo "duplex B=32 mid interleave=$i: $(line $O/h_b32_i$i.log)" | tee -a $O/h_summary.txt
done
for i in 0 1; do
  MMI_KV_INTERLEAVE=$i timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --kv-depth full --steps 20 --warmup 5 > $O/h_full_i$i.log 2>&1; echo "lm full context interleave=$i: $(line $O/h_full_i$i.log)" | tee -a $O/h_summary.txt
  MMI_KV_INTERLEAVE=$i timeout 240 python bench.py --no-cpu-baseline --no-extras --batch 64 --steps 40 --warmup 8 > $O/h_b64_i$i.log 2>&1; echo "duplex B=64 mid interleave=$i: $(line $O/h_b64_i$i.log)" | tee -a $O/h_summary.txt
done
