# last check of the round on the final tree: smoke, all GPU tests, the default benchmark line and its kernel trace on the same box
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
( time timeout 900 python bench.py ) > $O/bench_default2.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_default -o duplex -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/rocprof_default2.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py /tmp/prof_default/duplex_results.db --header "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (the default benchmark command: duplex, 32 sessions, 60 steps + 12 warm-up + 248 stagger steps)" > $O/r01_duplex_b32_final2_kernel_stats.csv
rocm-smi --showclocks > $O/rocm_smi_final2.log 2>&1
tail -n 3 $O/smoke.log; tail -n 3 $O/pytest_gpu.log; grep '"metric"' $O/bench_default2.log | cut -c1-330; head -8 $O/r01_duplex_b32_final2_kernel_stats.csv | cut -c1-140
