# round 2, call u: the measured tiling table of k_conv_wide on / off (MMI_CONV_NO_TUNE_TABLE), Mimi only at 32 / 64 / 8 sessions; Mimi GPU tests;
# default line
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
rm -f $O/ab_tune_table.txt
for rep in 1 2; do
  for b in 32 64 8; do
    VARS="MMI_CONV_NO_TUNE_TABLE=1" run ab_tune_table.txt "mimi only B=$b planner rule" --workload mimi --batch $b
    VARS="MMI_DUMMY=1" run ab_tune_table.txt "mimi only B=$b measured table" --workload mimi --batch $b
  done
done
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.log 2>&1
tail -3 $O/pytest_gpu_subset.log; cat $O/ab_tune_table.txt; grep '"metric"' $O/bench_default.log | cut -c1-300
