# Round-5 GPU call B: (1) engine-or-checker determinism at C5's shape, (2) the tests added since call A, (3) C5 / default bench lines
# with the new k_gemm_q8 tiling and its A/B, per-site tables at 64 sessions int8, (4) the counter passes.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do timeout 400 python tests/tools/c5_determinism_probe.py p$i 2>&1 | grep -E "^(ORACLE|ENGINE|REPEAT|VS_ORACLE)" | tee -a $O/c5_determinism.txt; done
timeout 400 python tests/tools/c5_determinism_probe.py bf16 --quant none 2>&1 | grep -E "^(ORACLE|ENGINE|REPEAT|VS_ORACLE)" | tee -a $O/c5_determinism.txt
MMI_NO_GRAPH=1 timeout 400 python tests/tools/c5_determinism_probe.py eager 2>&1 | grep -E "^(ENGINE|REPEAT)" | tee -a $O/c5_determinism.txt
timeout 900 python -m pytest tests/test_y_c5_int8_gpu.py tests/test_b_lm_gpu.py -m gpu -q -x --durations=12 -k "int8 or c5 or switch or long_ring or tiny_matches or state" > $O/pytest_b.log 2>&1; echo "pytest b rc=$?"; tail -22 $O/pytest_b.log | cut -c1-220
for v in grid serial; do MMI_Q8_TILES=$v timeout 300 python bench.py --no-cpu-baseline --no-extras --batch 64 --quant q8 > $O/bench_b64_q8_$v.log 2>&1; grep '"metric"' $O/bench_b64_q8_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b64 q8 tiles=$v ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))"; done
MMI_Q8_ACT=bf16 timeout 300 python bench.py --no-cpu-baseline --no-extras --batch 64 --quant q8 > $O/bench_b64_q8_wo.log 2>&1; grep '"metric"' $O/bench_b64_q8_wo.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b64 q8 weight-only ms/step %.3f' % d['ms_per_step'])"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_q8 -o q8 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --serial --batch 64 --quant q8 --launch-lists $O/launch_lists_q8 > $O/prof_q8.log 2>&1 ); echo "rocprof q8 rc=$?"
python scripts/rocpd_stats.py /tmp/prof_q8/q8_results.db --header "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras --serial --batch 64 --quant q8 (duplex, 64 sessions, int8 x int8, one stream)" > $O/r05_q8_b64_kernel_stats.csv
python scripts/rocpd_sites.py /tmp/prof_q8/q8_results.db $O/launch_lists_q8 --last 100 --header "per-site kernel time, serial schedule, 64 sessions, int8 weights x int8 activations, k_gemm_q8 one batch tile per workgroup" > $O/r05_q8_b64_sites.csv 2> $O/rocpd_sites_q8.err || head -5 $O/rocpd_sites_q8.err
grep "L\.\|dep\.\|TOTAL\|text" $O/r05_q8_b64_sites.csv | cut -c1-200 | head -30
bash scripts/gpu_pmc_step.sh
