# round 2, call p: split-K depth of the N = 4096 temporal GEMMs (out_proj, linear_out): 2 (default) against 3 and 4 workgroups per n-tile, LM only
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
run() { # file, label, bench args..., env via VARS
  local file=$1 label=$2; shift 2
  env $VARS timeout 200 python bench.py "$@" --no-cpu-baseline > $O/ab_tmp.log 2>&1
  echo "$label $(grep '"metric"' $O/ab_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))")" >> $O/$file
}
for rep in 1 2; do
  VARS="MMI_DUMMY=1" run ab_ksplit.txt "lm only ksplit=2 (default)" --workload lm
  VARS="MMI_GEMM_KSPLIT=3" run ab_ksplit.txt "lm only ksplit=3" --workload lm
  VARS="MMI_GEMM_KSPLIT=4" run ab_ksplit.txt "lm only ksplit=4" --workload lm
done
cat $O/ab_ksplit.txt
