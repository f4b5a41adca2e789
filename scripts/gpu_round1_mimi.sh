set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/box.log 2>&1
nproc >> gpurun_out/box.log; free -g | head -2 >> gpurun_out/box.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --workload mimi --batch 8 --steps 100 --warmup 20 > gpurun_out/bench_mimi_b8.log 2>&1
timeout 600 python bench.py --workload mimi --batch 32 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_mimi_b32.log 2>&1
timeout 600 python bench.py --workload mimi --batch 1 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_mimi_b1.log 2>&1
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_mimi_b32 -o mimi -- python $GRAFT_REPO_ROOT/bench.py --workload mimi --batch 32 --steps 30 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_mimi.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out | head -50
tail -5 gpurun_out/smoke.log gpurun_out/pytest_gpu.log gpurun_out/bench_mimi_b8.log gpurun_out/bench_mimi_b32.log gpurun_out/bench_mimi_b1.log
