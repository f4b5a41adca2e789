# round 2, first GPU call: the whole GPU suite (incl. the new full-depth / wide-batch / ring-wrap parity tests), the default
# benchmark line, its kernel trace joined with the launch lists (per-site table), the k_gemm_xlds staggered-tail A/B, C2 / C3 lines
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
nproc > $O/host.txt; free -g >> $O/host.txt; rocm-smi --showclocks >> $O/host.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -x -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
( time timeout 900 python bench.py ) > $O/bench_default.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_default -o duplex -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --launch-lists $O/ll > $O/rocprof_default.log 2>&1
cd $GRAFT_REPO_ROOT
HDR="rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (the default benchmark command: duplex, 32 sessions, 60 steps + 12 warm-up + 248 stagger steps)"
python scripts/rocpd_stats.py /tmp/prof_default/duplex_results.db --header "$HDR" > $O/r02_duplex_b32_a_kernel_stats.csv
python scripts/rocpd_sites.py /tmp/prof_default/duplex_results.db $O/ll --header "$HDR" > $O/r02_duplex_b32_a_sites.csv 2> $O/sites_err.log
python scripts/rocpd_step_trace.py /tmp/prof_default/duplex_results.db > $O/r02_duplex_b32_a_step_trace.csv 2>> $O/sites_err.log
MMI_TEST_XLDS_MODES=1,2 timeout 400 python -m pytest tests/test_zz_experimental_gpu.py -m gpu -q --timeout=300 > $O/xlds_mode2_parity.log 2>&1
for mode in 0 2 0 2; do
  MMI_GEMM_LDS=$mode timeout 200 python bench.py --no-cpu-baseline > $O/ab_lds$mode.log 2>&1
  echo "MMI_GEMM_LDS=$mode $(grep '"metric"' $O/ab_lds$mode.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f  dominant kernel %.2f us' % (d['ms_per_step'], 1e3*d['roofline']['avg_launch_ms']))")" >> $O/ab_lds.txt
done
timeout 200 python bench.py --workload lm --batch 1 --no-cpu-baseline > $O/bench_lm_b1.log 2>&1
timeout 200 python bench.py --workload mimi --batch 8 --no-cpu-baseline > $O/bench_mimi_b8.log 2>&1
timeout 200 python bench.py --workload lm --batch 32 --no-cpu-baseline > $O/bench_lm_b32.log 2>&1
timeout 200 python bench.py --workload mimi --batch 32 --no-cpu-baseline > $O/bench_mimi_b32.log 2>&1
tail -n 3 $O/smoke.log; tail -n 25 $O/pytest_gpu.log | cut -c1-200; grep '"metric"' $O/bench_default.log | cut -c1-1500; cat $O/ab_lds.txt; tail -3 $O/xlds_mode2_parity.log
for f in bench_lm_b1 bench_mimi_b8 bench_lm_b32 bench_mimi_b32; do grep '"metric"' $O/$f.log | cut -c1-260; done
head -40 $O/r02_duplex_b32_a_sites.csv
