# Round 4, GPU call A (gpurun --timeout 1500 -- 'bash scripts/gpu_r04_a.sh'): the new parity cases, the attention kernel A/B at
# both ring depths, the seeked default line with its extras, and the one-session (C3) per-site table under rocprofv3.
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
    print('ms/step %.3f p50 %.3f frames/s %.0f step-frac %.3f | norm1 %.1f in_proj %.1f attn %.1f out_proj %.1f norm2 %.1f ffn_in %.1f ffn_out %.1f' % (
        d['ms_per_step'], d.get('p50_ms_per_step', 0), d['value'], d.get('roofline', {}).get('step', {}).get('frac', 0),
        g('L.norm1'), g('L.in_proj'), g('L.attn'), g('L.out_proj'), g('L.norm2'), g('L.ffn_in'), g('L.ffn_out')))
except Exception as e:
    print('no line:', e)
PY
}
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
# ---- A/B first (short runs): attention kernel x ring depth, pipelined and LM alone
for depth in start mid; do for attn in split wave; do
  MMI_ATTN=$attn timeout 200 python bench.py --no-cpu-baseline --no-extras --kv-depth $depth --steps 40 --warmup 8 > $O/ab_${depth}_${attn}.log 2>&1
  echo "duplex depth=$depth attn=$attn: $(line $O/ab_${depth}_${attn}.log)" | tee -a $O/ab_summary.txt
done; done
for attn in split wave; do
  MMI_ATTN=$attn timeout 200 python bench.py --no-cpu-baseline --workload lm --batch 1 --steps 40 --warmup 8 > $O/ab_lm_b1_${attn}.log 2>&1
  echo "lm-only B=1 attn=$attn: $(line $O/ab_lm_b1_${attn}.log)" | tee -a $O/ab_summary.txt
  MMI_ATTN=$attn timeout 300 python bench.py --no-cpu-baseline --no-extras --workload lm --kv-depth full --steps 20 --warmup 5 > $O/ab_lm_full_${attn}.log 2>&1
  echo "lm-only full context attn=$attn: $(line $O/ab_lm_full_${attn}.log)" | tee -a $O/ab_summary.txt
done
# ---- the default line as the driver runs it (extras + the 32-layer CPU oracle)
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.log 2>&1; echo "default: $(line $O/bench_default.log)" | tee -a $O/ab_summary.txt
grep '"metric"' $O/bench_default.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('full_context', json.dumps(d.get('full_context'))[:400]); c3 = dict(d.get('c3') or {}); c3.pop('sites', None); print('c3', json.dumps(c3)[:500]); print('cpu_baseline', json.dumps(d.get('cpu_baseline'))[:900])" | tee -a $O/ab_summary.txt
# ---- new parity cases
timeout 1200 python -m pytest tests/test_mimi_gpu.py tests/test_dist_gpu.py tests/test_duplex_gpu.py -m gpu -q -x --timeout=900 -s > $O/pytest_new_a.log 2>&1; echo "pytest mimi/dist/duplex rc=$?"; grep -a "\[parity\]\|passed\|failed\|Error" $O/pytest_new_a.log | tail -12
timeout 1500 python -m pytest tests/test_lm_gpu.py -m gpu -q -x --timeout=900 -s -k "free_running or full_depth_32 or benchmark_ or ring_wrap" > $O/pytest_new_b.log 2>&1; echo "pytest lm rc=$?"; grep -a "\[parity\]\|passed\|failed\|Error" $O/pytest_new_b.log | grep -v "audio[1-6]" | tail -30
# ---- C3 under the tracer: kernel stats + per-site table of the one-session LM step
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_b1 -o lm -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload lm --batch 1 --steps 40 --warmup 8 --launch-lists $O/launch_lists_b1 > $O/rocprof_b1.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py /tmp/prof_b1/lm_results.db --header "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --workload lm --batch 1   (C3: Moshi-7B LMGen.step, one session)" > $O/r04_lm_b1_kernel_stats.csv
python scripts/rocpd_sites.py /tmp/prof_b1/lm_results.db $O/launch_lists_b1 --header "per-site kernel time, LMGen.step, ONE session (C3), ring 150 + deep" > $O/r04_lm_b1_sites.csv
cat $O/r04_lm_b1_sites.csv | head -30
