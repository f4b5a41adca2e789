# one iteration check: all GPU tests, then the three benches (duplex default, LM only, Mimi only) and a kernel trace
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout=600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 400 python bench.py --no-cpu-baseline > $O/bench_duplex_b32.log 2>&1
timeout 300 python bench.py --workload lm --batch 32 --no-cpu-baseline > $O/bench_lm_b32.log 2>&1
timeout 300 python bench.py --workload lm --batch 1 --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_lm_b1.log 2>&1
timeout 300 python bench.py --workload mimi --batch 32 --no-cpu-baseline > $O/bench_mimi_b32.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_iter -o duplex -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/rocprof_iter.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py /tmp/prof_iter/duplex_results.db > $O/iter_kernel_stats.csv 2>&1
tail -n 4 $O/pytest_gpu.log
for f in bench_duplex_b32 bench_lm_b32 bench_lm_b1 bench_mimi_b32; do echo $f; tail -n 1 $O/$f.log | cut -c1-240; done
head -14 $O/iter_kernel_stats.csv | cut -c1-160
