# round 2, call zb: three pack launches removed (producers also store the consuming linear's packed operand: latent -> RVQ input projection,
# RVQ gather -> output projection, decoder conv0 -> first transposed-conv GEMM) and the two commits of a step in one launch, against
# MMI_MIMI_PACK_LAUNCHES=1 MMI_MIMI_TWO_COMMITS=1; Mimi + batcher GPU tests; default line
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_mimi_gpu.py tests/test_batcher_gpu.py -m gpu -q --timeout=600 > $O/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_subset.log
run() { # file, label, bench args..., env via VARS
  local file=$1 label=$2; shift 2
  env $VARS timeout 200 python bench.py "$@" --no-cpu-baseline > $O/ab_tmp.log 2>&1
  echo "$label $(grep '"metric"' $O/ab_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))")" >> $O/$file
}
rm -f $O/ab_fewer_launches.txt
for rep in 1 2; do
  VARS="MMI_MIMI_PACK_LAUNCHES=1 MMI_MIMI_TWO_COMMITS=1" run ab_fewer_launches.txt "mimi only B=32 pack + two commit launches" --workload mimi
  VARS="MMI_DUMMY=1" run ab_fewer_launches.txt "mimi only B=32 producers pack, one commit launch" --workload mimi
done
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.log 2>&1
tail -3 $O/pytest_gpu_subset.log; cat $O/ab_fewer_launches.txt; grep '"metric"' $O/bench_default.log | cut -c1-300
