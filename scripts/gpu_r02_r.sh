# round 2, call r: the whole GPU suite on the tree, smoke, default line (with the CPU baseline leg), kernel trace + sites
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 600 python bench.py > $O/bench_default_full.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_default -o duplex -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --launch-lists $O/ll > $O/rocprof_default.log 2>&1
cd $GRAFT_REPO_ROOT
HDR="rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (the default benchmark command: duplex, 32 sessions, 60 steps + 12 warm-up + 248 stagger steps)"
python scripts/rocpd_stats.py /tmp/prof_default/duplex_results.db --header "$HDR" > $O/r02_duplex_b32_r_kernel_stats.csv
python scripts/rocpd_sites.py /tmp/prof_default/duplex_results.db $O/ll --header "$HDR" > $O/r02_duplex_b32_r_sites.csv 2> $O/sites_err.log
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3; tail -2 $O/smoke.log; grep '"metric"' $O/bench_default_full.log | cut -c1-400; grep -E "TOTAL" $O/r02_duplex_b32_r_sites.csv
