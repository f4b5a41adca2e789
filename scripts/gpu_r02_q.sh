# round 2, call q: the generic conv epilogue with its launch-uniform flags hoisted out of the per-value loop, against the engine of
# commit 873a662 (Mimi only, 32 sessions); Mimi GPU tests
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
for rep in 1 2 3; do
  VARS="MMI_LIB_PATH=$GRAFT_REPO_ROOT/ab_old/libmoshi_mi_873a662.so" run ab_epilogue.txt "mimi only B=32 873a662 (flags tested per value)" --workload mimi
  VARS="MMI_DUMMY=1" run ab_epilogue.txt "mimi only B=32 flags hoisted" --workload mimi
done
tail -3 $O/pytest_gpu_subset.log; cat $O/ab_epilogue.txt
