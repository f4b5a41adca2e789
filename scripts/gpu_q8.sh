set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_lm_gpu.py -m gpu -q --timeout=600 > gpurun_out/pytest_lm_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_lm_gpu.log
timeout 300 python bench.py --workload lm --batch 32 --steps 60 --warmup 12 --no-cpu-baseline --quant q8 > gpurun_out/bench_lm_b32_q8.log 2>&1
timeout 300 python bench.py --workload lm --batch 64 --steps 60 --warmup 12 --no-cpu-baseline --quant q8 > gpurun_out/bench_lm_b64_q8.log 2>&1
timeout 300 python bench.py --workload lm --batch 1 --steps 100 --warmup 20 --no-cpu-baseline --quant q8 > gpurun_out/bench_lm_b1_q8.log 2>&1
timeout 400 python bench.py --batch 64 --no-cpu-baseline --quant q8 > gpurun_out/bench_duplex_b64_q8.log 2>&1
timeout 400 python bench.py --no-cpu-baseline --quant q8 > gpurun_out/bench_duplex_b32_q8.log 2>&1
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/bench_duplex_b32.log 2>&1
tail -n 4 gpurun_out/pytest_lm_gpu.log; for f in bench_lm_b32_q8 bench_lm_b64_q8 bench_lm_b1_q8 bench_duplex_b64_q8 bench_duplex_b32_q8 bench_duplex_b32; do echo $f; tail -n 1 gpurun_out/$f.log | cut -c1-330; done
