# iteration check of this session: all GPU tests (incl. batcher + fp8), default bench, fp8 / q8 benches at the C5 batch
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 400 python bench.py --batch 64 --no-cpu-baseline --quant fp8 > $O/bench_duplex_b64_fp8.log 2>&1
timeout 400 python bench.py --batch 64 --no-cpu-baseline --quant q8 > $O/bench_duplex_b64_q8.log 2>&1
tail -n 6 $O/pytest_gpu.log
for f in bench_duplex_b64_fp8 bench_duplex_b64_q8; do echo $f; tail -n 1 $O/$f.log | cut -c1-420; done
