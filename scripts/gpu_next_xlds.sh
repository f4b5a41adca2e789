# First GPU call of the next round for k_gemm_xlds: whole-kernel microbenchmark of both tails at 32 / 64 sessions,
# parity of the staggered tail on hardware, and a same-box A/B of the default benchmark with MMI_GEMM_LDS = 0 / 1 / 2.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -Imoshi_amd/csrc scripts/gemm_microbench.hip moshi_amd/csrc/api_common.hip -o /tmp/gemm_microbench > $O/mb_build.log 2>&1 || cat $O/mb_build.log
for B in 32 64; do MB_LDS=1 timeout 200 /tmp/gemm_microbench $B 1 > $O/next_mb_lds_b$B.txt 2>&1; done
MMI_TEST_XLDS_MODES=1,2 timeout 400 python -m pytest tests/test_zz_experimental_gpu.py -m gpu -q --timeout=300 > $O/next_xlds_mode2_parity.log 2>&1
for mode in 0 1 2 0 1 2; do
  MMI_GEMM_LDS=$mode timeout 200 python bench.py --no-cpu-baseline > $O/next_bench_lds$mode.log 2>&1
  echo "MMI_GEMM_LDS=$mode $(grep '"metric"' $O/next_bench_lds$mode.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f  dominant kernel %.2f us' % (d['ms_per_step'], 1e3*d['roofline']['avg_launch_ms']))")"
done
for qn in q8 fp8; do for mode in 0 1; do
  MMI_GEMM_LDS=$mode timeout 200 python bench.py --batch 64 --quant $qn --no-cpu-baseline > $O/next_bench_b64_${qn}_lds$mode.log 2>&1
  echo "64 sessions $qn MMI_GEMM_LDS=$mode $(grep '"metric"' $O/next_bench_b64_${qn}_lds$mode.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f' % d['ms_per_step'])")"
done; done
grep -E "^==|k_gemm_xlds|prototype|main loop|32x[12] ntw1 w8 u[24]" $O/next_mb_lds_b32.txt $O/next_mb_lds_b64.txt | cut -c1-140
cat $O/next_xlds_mode2_parity.log | tail -3
