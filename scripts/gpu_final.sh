set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_default -o duplex -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_default.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py /tmp/prof_default/duplex_results.db --header "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (the default benchmark command: duplex, 32 sessions, 60 steps + 12 warm-up + 248 stagger steps)" > gpurun_out/r01_duplex_b32_default_kernel_stats.csv
tail -n 3 gpurun_out/smoke.log; tail -n 4 gpurun_out/pytest_gpu.log; tail -n 6 gpurun_out/bench_default.log | cut -c1-1500; head -8 gpurun_out/r01_duplex_b32_default_kernel_stats.csv | cut -c1-150
