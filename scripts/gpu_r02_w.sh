# round 2, call w: the 64-channel residual blocks on the two-launch path under every tiling, against k_resblock (30.4 / 33.5 us in call s)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rm -f $O/res64_sweep.txt
for cfg in "0 0" "2 1" "2 2" "2 4" "2 8" "1 1" "1 2" "1 4" "1 8"; do
  set -- $cfg
  if [ "$1" = "0" ]; then VARS="MMI_MIMI_NO_RES_FUSION=1"; else VARS="MMI_MIMI_NO_RES_FUSION=1 MMI_CONV_MTB=$1 MMI_CONV_W=$2"; fi
  cd /tmp && env $VARS timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o mimi -- python $GRAFT_REPO_ROOT/bench.py --workload mimi --no-cpu-baseline --launch-lists $O/ll_x > $O/rocprof_x.log 2>&1
  cd $GRAFT_REPO_ROOT
  echo "[two launches, MTB=$1 W=$2] $(python scripts/rocpd_sites.py /tmp/prof_x/mimi_results.db $O/ll_x --header x 2>> $O/sites_err.log | grep -E 'enc.res0|dec.res3|enc.conv0|TOTAL' | awk -F, '{printf "%s=%s ", $2, $4}')" >> $O/res64_sweep.txt
  rm -rf /tmp/prof_x
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o mimi -- python $GRAFT_REPO_ROOT/bench.py --workload mimi --no-cpu-baseline --launch-lists $O/ll_x > $O/rocprof_x.log 2>&1
cd $GRAFT_REPO_ROOT
echo "[k_resblock] $(python scripts/rocpd_sites.py /tmp/prof_x/mimi_results.db $O/ll_x --header x 2>> $O/sites_err.log | grep -E 'enc.res0|dec.res3|enc.conv0|TOTAL' | awk -F, '{printf "%s=%s ", $2, $4}')" >> $O/res64_sweep.txt
cat $O/res64_sweep.txt
