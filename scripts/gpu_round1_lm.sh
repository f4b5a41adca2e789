set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/box.log 2>&1
nproc >> gpurun_out/box.log; free -g | head -2 >> gpurun_out/box.log
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 700 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python bench.py --workload lm --batch 1 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_lm_b1.log 2>&1
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/bench_duplex_b32.log 2>&1
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_duplex_b32 -o duplex -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --stagger 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_duplex.log 2>&1
cd $GRAFT_REPO_ROOT
for f in smoke pytest_gpu bench_lm_b1 bench_duplex_b32; do echo "== $f"; tail -n 8 gpurun_out/$f.log; done
