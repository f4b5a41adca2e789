# round 2, call t (8 and 64 sessions): tiling sweep of k_conv_wide (m-tiles per wave x split-K waves, forced on every audio-rate layer at once) - per-site times
# of the Mimi-only step, to see whether the planner's per-layer choices are the best ones; text sampler with / without cached logits
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
for BB in 8 64; do
rm -f $O/conv_sweep_b$BB.txt
for cfg in "0 0" "4 1" "4 2" "4 4" "2 1" "2 2" "2 4" "2 8" "1 1" "1 2" "1 4" "1 8"; do
  set -- $cfg
  if [ "$1" = "0" ]; then VARS="MMI_DUMMY=1"; else VARS="MMI_CONV_MTB=$1 MMI_CONV_W=$2"; fi
  cd /tmp && env $VARS timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$1_$2 -o mimi -- python $GRAFT_REPO_ROOT/bench.py --workload mimi --batch $BB --no-cpu-baseline --launch-lists $O/ll_$1_$2 > $O/rocprof_$1_$2.log 2>&1
  cd $GRAFT_REPO_ROOT
  echo "[MTB=$1 W=$2] $(python scripts/rocpd_sites.py /tmp/prof_$1_$2/mimi_results.db $O/ll_$1_$2 --header x 2>> $O/sites_err.log | grep -E 'conv0|res[0-3]|down[0-3]|convtr[0-3]|dec.final|TOTAL' | awk -F, '{printf "%s=%s ", $2, $4}')" >> $O/conv_sweep_b$BB.txt
  rm -rf /tmp/prof_$1_$2
done
done
cat $O/conv_sweep_b8.txt $O/conv_sweep_b64.txt
