# round 2, last call: smoke() and the default line (without the CPU-baseline leg) on the final tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 30 python bench.py --no-cpu-baseline > $O/bench_default_last.log 2>&1
timeout 25 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_last.log 2>&1; echo "smoke rc=$?" >> $O/smoke_last.log
grep '"metric"' $O/bench_default_last.log | cut -c1-300; tail -2 $O/smoke_last.log
