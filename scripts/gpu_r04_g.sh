# Round 4, GPU call G: C5 as SURVEY 8d states it (64 sessions, 1-byte weights AND the fp8 KV ring), the fp8-ring attention A/B, and the
# one-session (C3) tables again with the round's final attention kernel.
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
rm -f $O/g_summary.txt
for q in q8 fp8 none; do
  timeout 240 python bench.py --no-cpu-baseline --no-extras --batch 64 --quant $q --kv fp8 --steps 40 --warmup 8 > $O/g_b64_${q}_kv8.log 2>&1; echo "b64 $q + fp8 KV ring: $(line $O/g_b64_${q}_kv8.log)" | tee -a $O/g_summary.txt
done
MMI_Q8_ACT=bf16 timeout 240 python bench.py --no-cpu-baseline --no-extras --batch 64 --quant q8 --kv fp8 --steps 40 --warmup 8 > $O/g_b64_q8w_kv8.log 2>&1; echo "b64 q8 weight-only + fp8 KV ring: $(line $O/g_b64_q8w_kv8.log)" | tee -a $O/g_summary.txt
MMI_ATTN=split timeout 240 python bench.py --no-cpu-baseline --no-extras --batch 64 --quant q8 --kv fp8 --steps 40 --warmup 8 > $O/g_b64_q8_kv8_split.log 2>&1; echo "b64 q8 + fp8 KV ring, chunked attention: $(line $O/g_b64_q8_kv8_split.log)" | tee -a $O/g_summary.txt
timeout 240 python bench.py --no-cpu-baseline --no-extras --kv fp8 --steps 40 --warmup 8 > $O/g_b32_kv8.log 2>&1; echo "b32 bf16 + fp8 KV ring: $(line $O/g_b32_kv8.log)" | tee -a $O/g_summary.txt
timeout 300 python -m pytest tests/test_lm_gpu.py -m gpu -q -x --timeout=280 -k "fp8_kv" > $O/pytest_kv8.log 2>&1; echo "pytest fp8 KV rc=$?" | tee -a $O/g_summary.txt
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_b1 -o lm -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload lm --batch 1 --steps 40 --warmup 8 --launch-lists $O/launch_lists_b1 > $O/rocprof_b1.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py /tmp/prof_b1/lm_results.db --header "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --workload lm --batch 1   (C3: Moshi-7B LMGen.step, one session)" > $O/r04_lm_b1_kernel_stats.csv
python scripts/rocpd_sites.py /tmp/prof_b1/lm_results.db $O/launch_lists_b1 --header "per-site kernel time, LMGen.step, ONE session (C3), ring 150 + deep" > $O/r04_lm_b1_sites.csv
grep "^lm" $O/r04_lm_b1_sites.csv | head -24
