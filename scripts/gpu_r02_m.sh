# round 2, call m: k_resblock with its own branch-free epilogue - Mimi GPU tests, same-box A/B against two launches per block, per-site times
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_mimi_gpu.py -m gpu -q --timeout=600 > $O/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_subset.log
run() { # file, label, bench args..., env via VARS
  local file=$1 label=$2; shift 2
  env $VARS timeout 200 python bench.py "$@" --no-cpu-baseline > $O/ab_tmp.log 2>&1
  echo "$label $(grep '"metric"' $O/ab_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))")" >> $O/$file
}
for rep in 1 2; do
  VARS="MMI_MIMI_NO_RES_FUSION=1" run ab_resblock.txt "mimi only B=32 two launches per block" --workload mimi
  VARS="MMI_DUMMY=1" run ab_resblock.txt "mimi only B=32 k_resblock" --workload mimi
done
for dbg in 0 8; do
  cd /tmp && MMI_RES_DBG=$dbg timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$dbg -o mimi -- python $GRAFT_REPO_ROOT/bench.py --workload mimi --no-cpu-baseline --launch-lists $O/ll_$dbg > $O/rocprof_$dbg.log 2>&1
  cd $GRAFT_REPO_ROOT
  python scripts/rocpd_sites.py /tmp/prof_$dbg/mimi_results.db $O/ll_$dbg --header "MMI_RES_DBG=$dbg" 2>> $O/sites_err.log | grep -E "res[0-3]|conv0|TOTAL" | tr '\n' ' ' >> $O/resblock_ablation.txt
  echo "  [dbg=$dbg]" >> $O/resblock_ablation.txt
done
tail -3 $O/pytest_gpu_subset.log; cat $O/ab_resblock.txt $O/resblock_ablation.txt
