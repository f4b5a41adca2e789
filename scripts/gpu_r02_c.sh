# round 2, third GPU call: new GPU tests (step hooks, cross-attention, depformer attention rewrite) + full suite, A/B of the
# depformer attention kernels, kernel trace + per-site table
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export MMI_DEP_ATTN_OLD=1; else unset MMI_DEP_ATTN_OLD; fi
  timeout 200 python bench.py --workload lm --no-cpu-baseline > $O/ab_depattn_old$v.log 2>&1
  echo "lm only MMI_DEP_ATTN_OLD=$v $(grep '"metric"' $O/ab_depattn_old$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))")" >> $O/ab_depattn.txt
done
unset MMI_DEP_ATTN_OLD
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_default -o duplex -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --launch-lists $O/ll > $O/rocprof_default.log 2>&1
cd $GRAFT_REPO_ROOT
HDR="rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (the default benchmark command: duplex, 32 sessions, 60 steps + 12 warm-up + 248 stagger steps)"
python scripts/rocpd_stats.py /tmp/prof_default/duplex_results.db --header "$HDR" > $O/r02_duplex_b32_c_kernel_stats.csv
python scripts/rocpd_sites.py /tmp/prof_default/duplex_results.db $O/ll --header "$HDR" > $O/r02_duplex_b32_c_sites.csv 2> $O/sites_err.log
tail -n 3 $O/smoke.log; grep -E "passed|failed|Error|error" $O/pytest_gpu.log | cut -c1-300 | tail -12; cat $O/ab_depattn.txt
grep -E "^lm," $O/r02_duplex_b32_c_sites.csv
