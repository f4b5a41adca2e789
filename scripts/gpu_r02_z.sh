# round 2, final call: the whole GPU suite on the final tree, smoke, the default line (with its CPU-baseline leg), kernel trace + per-site
# table of the same command, and the other configurations' lines for the record (C2 Mimi 8 sessions, C3 LM 1 session, 64 sessions bf16 / int8 / fp8, served)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 600 python bench.py > $O/bench_default_full.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_default -o duplex -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --launch-lists $O/ll > $O/rocprof_default.log 2>&1
cd $GRAFT_REPO_ROOT
HDR="rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (the default benchmark command: duplex, 32 sessions, 60 steps + 12 warm-up + 248 stagger steps)"
python scripts/rocpd_stats.py /tmp/prof_default/duplex_results.db --header "$HDR" > $O/r02_duplex_b32_z_kernel_stats.csv
python scripts/rocpd_sites.py /tmp/prof_default/duplex_results.db $O/ll --header "$HDR" > $O/r02_duplex_b32_z_sites.csv 2> $O/sites_err.log
timeout 200 python bench.py --workload mimi --batch 8 --no-cpu-baseline > $O/bench_mimi_b8.log 2>&1
timeout 200 python bench.py --workload mimi --no-cpu-baseline > $O/bench_mimi_b32.log 2>&1
timeout 200 python bench.py --workload lm --batch 1 --no-cpu-baseline > $O/bench_lm_b1.log 2>&1
timeout 200 python bench.py --workload lm --no-cpu-baseline > $O/bench_lm_b32.log 2>&1
timeout 300 python bench.py --batch 64 --no-cpu-baseline > $O/bench_duplex_b64_bf16.log 2>&1
timeout 300 python bench.py --batch 64 --quant q8 --no-cpu-baseline > $O/bench_duplex_b64_q8.log 2>&1
timeout 300 python bench.py --batch 64 --quant fp8 --no-cpu-baseline > $O/bench_duplex_b64_fp8.log 2>&1
timeout 300 python bench.py --workload served --no-cpu-baseline > $O/bench_served_b32.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3; tail -2 $O/smoke.log; grep '"metric"' $O/bench_default_full.log | cut -c1-330
for f in bench_mimi_b8 bench_mimi_b32 bench_lm_b1 bench_lm_b32 bench_duplex_b64_bf16 bench_duplex_b64_q8 bench_duplex_b64_fp8 bench_served_b32; do echo "$f $(grep '"metric"' $O/$f.log | cut -c1-200)"; done
grep -E "TOTAL" $O/r02_duplex_b32_z_sites.csv
