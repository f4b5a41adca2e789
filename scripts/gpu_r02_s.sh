# round 2, call s: tiling sweep of k_conv_wide (m-tiles per wave x split-K waves, forced on every audio-rate layer at once) - per-site times
# of the Mimi-only step, to see whether the planner's per-layer choices are the best ones; text sampler with / without cached logits
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rm -f $O/conv_sweep.txt
for cfg in "0 0" "4 1" "4 2" "4 4" "2 1" "2 2" "2 4" "2 8" "1 1" "1 2" "1 4" "1 8"; do
  set -- $cfg
  if [ "$1" = "0" ]; then VARS="MMI_DUMMY=1"; else VARS="MMI_CONV_MTB=$1 MMI_CONV_W=$2"; fi
  cd /tmp && env $VARS timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$1_$2 -o mimi -- python $GRAFT_REPO_ROOT/bench.py --workload mimi --no-cpu-baseline --launch-lists $O/ll_$1_$2 > $O/rocprof_$1_$2.log 2>&1
  cd $GRAFT_REPO_ROOT
  echo "[MTB=$1 W=$2] $(python scripts/rocpd_sites.py /tmp/prof_$1_$2/mimi_results.db $O/ll_$1_$2 --header x 2>> $O/sites_err.log | grep -E 'conv0|res[0-3]|down[0-3]|convtr[0-3]|dec.final|TOTAL' | awk -F, '{printf "%s=%s ", $2, $4}')" >> $O/conv_sweep.txt
  rm -rf /tmp/prof_$1_$2
done
run() { # file, label, bench args..., env via VARS
  local file=$1 label=$2; shift 2
  env $VARS timeout 200 python bench.py "$@" --no-cpu-baseline > $O/ab_tmp.log 2>&1
  echo "$label $(grep '"metric"' $O/ab_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))")" >> $O/$file
}
rm -f $O/ab_text_sampler.txt
for rep in 1 2; do
  VARS="MMI_SAMPLE_TEXT_NOCACHE=1" run ab_text_sampler.txt "lm only text sampler re-reads its logits" --workload lm
  VARS="MMI_DUMMY=1" run ab_text_sampler.txt "lm only text sampler keeps them in registers" --workload lm
done
cat $O/conv_sweep.txt $O/ab_text_sampler.txt
