set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 400 python bench.py --workload lm --warmup 3000 --stagger 0 --steps 40 --no-cpu-baseline > $O/bench_lm_b32_fullctx_kvbf16.log 2>&1
timeout 400 python bench.py --workload lm --warmup 3000 --stagger 0 --steps 40 --no-cpu-baseline --kv fp8 > $O/bench_lm_b32_fullctx_kvfp8.log 2>&1
grep -E "passed|failed|FAILED|Error|engine:" $O/pytest_gpu.log | tail -14
for f in bench_lm_b32_fullctx_kvbf16 bench_lm_b32_fullctx_kvfp8; do echo $f; tail -n 1 $O/$f.log | cut -c1-330; done
