cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
MMI_BENCH_TRACE=1 timeout 120 python -X faulthandler bench.py --no-cpu-baseline > $O/n_bench.log 2>&1; echo rc=$?
grep -v '"metric"' $O/n_bench.log | tail -30
