# round 2, call l: where k_resblock's time goes - timing ablations (MMI_RES_DBG bits: 1 no stage-0 loads, 2 no stage-1 MFMAs, 4 no stage 2,
# 8 no epilogue stores), Mimi only, per-site times from the kernel trace
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
for dbg in 0 1 2 4 8 6 7 15; do
  cd /tmp && MMI_RES_DBG=$dbg timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$dbg -o mimi -- python $GRAFT_REPO_ROOT/bench.py --workload mimi --no-cpu-baseline --launch-lists $O/ll_$dbg > $O/rocprof_$dbg.log 2>&1
  cd $GRAFT_REPO_ROOT
  python scripts/rocpd_sites.py /tmp/prof_$dbg/mimi_results.db $O/ll_$dbg --header "MMI_RES_DBG=$dbg" 2>> $O/sites_err.log | grep -E "res[0-3]|TOTAL" | tr '\n' ' ' >> $O/resblock_ablation.txt
  echo "  [dbg=$dbg]" >> $O/resblock_ablation.txt
done
cat $O/resblock_ablation.txt
