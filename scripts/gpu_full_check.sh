set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python bench.py --workload mimi --batch 32 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_mimi_b32.log 2>&1
timeout 300 python bench.py --workload mimi --batch 8 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_mimi_b8.log 2>&1
timeout 300 python bench.py --workload mimi --batch 1 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_mimi_b1.log 2>&1
timeout 300 python bench.py --workload lm --batch 32 --steps 60 --warmup 12 --no-cpu-baseline > gpurun_out/bench_lm_b32.log 2>&1
timeout 300 python bench.py --workload lm --batch 1 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_lm_b1.log 2>&1
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/bench_duplex_b32.log 2>&1
timeout 400 python bench.py --batch 64 --no-cpu-baseline > gpurun_out/bench_duplex_b64.log 2>&1
timeout 400 python bench.py --batch 8 --no-cpu-baseline > gpurun_out/bench_duplex_b8.log 2>&1
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_duplex_b32 -o duplex -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --stagger 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_duplex.log 2>&1
cd $GRAFT_REPO_ROOT
tail -n 5 gpurun_out/pytest_gpu.log; for f in bench_mimi_b32 bench_mimi_b8 bench_mimi_b1 bench_lm_b32 bench_lm_b1 bench_duplex_b32 bench_duplex_b64 bench_duplex_b8; do echo $f; tail -n 1 gpurun_out/$f.log | cut -c1-250; done
