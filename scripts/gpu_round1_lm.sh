set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 2400 python -m pytest tests -m gpu -q --timeout=1200 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench_duplex_b32.log 2>&1
timeout 600 python bench.py --workload lm --batch 1 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_lm_b1.log 2>&1
timeout 600 python bench.py --workload lm --batch 32 --steps 60 --warmup 12 --no-cpu-baseline > gpurun_out/bench_lm_b32.log 2>&1
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_duplex_b32 -o duplex -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --stagger 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_duplex.log 2>&1
cd $GRAFT_REPO_ROOT
for f in smoke pytest_gpu bench_duplex_b32 bench_lm_b1 bench_lm_b32; do echo "== $f"; tail -n 6 gpurun_out/$f.log; done
