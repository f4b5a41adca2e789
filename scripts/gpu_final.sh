# round-end check: smoke, all GPU tests, the default benchmark line (with cpu_baseline), its rocprofv3 kernel trace,
# the served (PCIe-inclusive) rate and the C5 (fp8, 64 sessions) line
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 900 python -m pytest tests -m gpu -q --timeout=600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
( time timeout 900 python bench.py ) > $O/bench_default.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_default -o duplex -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/rocprof_default.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py /tmp/prof_default/duplex_results.db --header "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (the default benchmark command: duplex, 32 sessions, 60 steps + 12 warm-up + 248 stagger steps)" > $O/r01_duplex_b32_final_kernel_stats.csv
timeout 400 python bench.py --workload served --no-cpu-baseline > $O/bench_served_b32.log 2>&1
timeout 400 python bench.py --batch 64 --no-cpu-baseline --quant fp8 --kv fp8 --warmup 3000 --stagger 0 --steps 40 > $O/bench_duplex_b64_fp8_kvfp8_fullctx.log 2>&1
MMI_GEMM_NTW=2 timeout 400 python bench.py --batch 64 --no-cpu-baseline --quant q8 > $O/bench_duplex_b64_q8_ntw2.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_b64 -o duplex -- python $GRAFT_REPO_ROOT/bench.py --batch 64 --quant q8 --no-cpu-baseline > $O/rocprof_b64_q8.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py /tmp/prof_b64/duplex_results.db --header "rocprofv3 --kernel-trace --stats -- python bench.py --batch 64 --quant q8 --no-cpu-baseline   (C5 weight format and batch: 64 sessions, int8 linears)" > $O/r01_duplex_b64_q8_kernel_stats.csv
tail -n 3 $O/smoke.log; tail -n 6 $O/pytest_gpu.log; tail -n 6 $O/bench_default.log | cut -c1-1800; tail -n 1 $O/bench_served_b32.log | cut -c1-700; tail -n 1 $O/bench_duplex_b64_fp8_kvfp8_fullctx.log | cut -c1-400; tail -n 1 $O/bench_duplex_b64_q8_ntw2.log | cut -c1-300; head -8 $O/r01_duplex_b64_q8_kernel_stats.csv | cut -c1-150; head -10 $O/r01_duplex_b32_final_kernel_stats.csv | cut -c1-150
