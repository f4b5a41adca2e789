# round 2, second GPU call: GPU suite on the new defaults (ELU hoisting, sampler fast path, k_gemm_xlds default), then same-box
# A/Bs of the default benchmark: ELU hoisting off / on, and the kernel trace + per-site table of the new default
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export MMI_NO_ELU_HOIST=1; else unset MMI_NO_ELU_HOIST; fi
  timeout 200 python bench.py --no-cpu-baseline > $O/ab_elu_nohoist$v.log 2>&1
  echo "MMI_NO_ELU_HOIST=$v $(grep '"metric"' $O/ab_elu_nohoist$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))")" >> $O/ab_elu.txt
done
unset MMI_NO_ELU_HOIST
for v in 1 0; do
  if [ $v = 1 ]; then export MMI_NO_ELU_HOIST=1; else unset MMI_NO_ELU_HOIST; fi
  timeout 200 python bench.py --workload mimi --batch 32 --no-cpu-baseline > $O/ab_mimi_nohoist$v.log 2>&1
  echo "mimi only MMI_NO_ELU_HOIST=$v $(grep '"metric"' $O/ab_mimi_nohoist$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f' % d['ms_per_step'])")" >> $O/ab_elu.txt
done
unset MMI_NO_ELU_HOIST
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_default -o duplex -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --launch-lists $O/ll > $O/rocprof_default.log 2>&1
cd $GRAFT_REPO_ROOT
HDR="rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (the default benchmark command: duplex, 32 sessions, 60 steps + 12 warm-up + 248 stagger steps)"
python scripts/rocpd_stats.py /tmp/prof_default/duplex_results.db --header "$HDR" > $O/r02_duplex_b32_b_kernel_stats.csv
python scripts/rocpd_sites.py /tmp/prof_default/duplex_results.db $O/ll --header "$HDR" > $O/r02_duplex_b32_b_sites.csv 2> $O/sites_err.log
tail -n 3 $O/smoke.log; grep -E "parity\]|passed|failed" $O/pytest_gpu.log | cut -c1-200; cat $O/ab_elu.txt
grep -E "sample|enc\.|dec\.|TOTAL" $O/r02_duplex_b32_b_sites.csv
