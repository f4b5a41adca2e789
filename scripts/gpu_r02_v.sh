# round 2, call v: sweep of the small-GEMM planner (k_gemm_f32: split-K depth forced on every natural-output GEMM, n-subtile spreading off),
# per-site times of the Mimi-only step at 32 sessions
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rm -f $O/gemm_f32_sweep.txt
for cfg in "default" "MMI_CONV_KSPLIT=1" "MMI_CONV_KSPLIT=2" "MMI_CONV_KSPLIT=4" "MMI_CONV_KSPLIT=8" "MMI_CONV_NO_SPREAD=1"; do
  if [ "$cfg" = "default" ]; then VARS="MMI_DUMMY=1"; else VARS="$cfg"; fi
  cd /tmp && env $VARS timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o mimi -- python $GRAFT_REPO_ROOT/bench.py --workload mimi --no-cpu-baseline --launch-lists $O/ll_x > $O/rocprof_x.log 2>&1
  cd $GRAFT_REPO_ROOT
  echo "[$cfg] $(python scripts/rocpd_sites.py /tmp/prof_x/mimi_results.db $O/ll_x --header x 2>> $O/sites_err.log | grep -E 'enc.down3|enc.final|enc.tr|enc.downsample|enc.rvq|dec.dequant|dec.tr|dec.conv0|dec.convtr0|TOTAL' | awk -F, '{printf "%s=%s(%s) ", $2, $4, $3}')" >> $O/gemm_f32_sweep.txt
  rm -rf /tmp/prof_x
done
cat $O/gemm_f32_sweep.txt
