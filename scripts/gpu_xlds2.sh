set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
MMI_GEMM_LDS=1 timeout 120 python bench.py --no-cpu-baseline > $O/xlds2_on_b32.log 2>&1
timeout 120 python bench.py --no-cpu-baseline > $O/xlds2_off_b32.log 2>&1
MMI_GEMM_LDS=1 timeout 120 python bench.py --batch 64 --no-cpu-baseline > $O/xlds2_on_b64.log 2>&1
for f in xlds2_on_b32 xlds2_off_b32 xlds2_on_b64; do echo $f $(grep '"metric"' $O/$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('ms/step %.3f p50 %.3f frames/s %.0f | dominant kernel %.2f us' % (d['ms_per_step'], d['p50_ms_per_step'], d['value'], 1e3*r.get('avg_launch_ms',0)))"); done
