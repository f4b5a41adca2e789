# round 2, call o: the 32-row tile (k_gemm_xlds, octet sharing) at small batches: LM only at 1 / 8 / 16 sessions, T = 16 (default) against T = 32
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
for b in 1 8 16; do
  VARS="MMI_DUMMY=1" run ab_tile.txt "lm only B=$b T=16" --workload lm --batch $b
  VARS="MMI_LM_TILE=32" run ab_tile.txt "lm only B=$b T=32" --workload lm --batch $b
  VARS="MMI_LM_TILE=32 MMI_GEMM_LDS=0" run ab_tile.txt "lm only B=$b T=32 no xlds" --workload lm --batch $b
done
VARS="MMI_DUMMY=1" run ab_tile.txt "duplex B=16 T=16"  --batch 16
VARS="MMI_LM_TILE=32" run ab_tile.txt "duplex B=16 T=32" --batch 16
cat $O/ab_tile.txt
