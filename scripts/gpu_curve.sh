# one box: the session-count curve of the duplex step (bf16), the single-session LM latency (C3), Mimi alone at 8 (C2), and the
# 64-session int8 step with one / two n-tiles per workgroup
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
for B in 1 8 16 32 48 64; do
  timeout 300 python bench.py --batch $B --no-cpu-baseline > $O/curve_duplex_b$B.log 2>&1
done
timeout 300 python bench.py --workload lm --batch 1 --steps 100 --warmup 20 --no-cpu-baseline > $O/curve_lm_b1.log 2>&1
timeout 300 python bench.py --workload mimi --batch 8 --steps 200 --warmup 20 --no-cpu-baseline > $O/curve_mimi_b8.log 2>&1
timeout 300 python bench.py --batch 64 --quant q8 --no-cpu-baseline > $O/curve_duplex_b64_q8.log 2>&1
MMI_GEMM_NTW=2 timeout 300 python bench.py --batch 64 --quant q8 --no-cpu-baseline > $O/curve_duplex_b64_q8_ntw2.log 2>&1
timeout 300 python bench.py --batch 32 --quant q8 --no-cpu-baseline > $O/curve_duplex_b32_q8.log 2>&1
timeout 300 python bench.py --batch 32 --quant fp8 --no-cpu-baseline > $O/curve_duplex_b32_fp8.log 2>&1
for f in $O/curve_*.log; do echo $(basename $f) $(grep '"metric"' $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f p95 %.3f frames/s %.0f' % (d['ms_per_step'], d['p50_ms_per_step'], d['p95_ms_per_step'], d['value']))"); done
