# round 2, call k (k_resblock with the input window staged in LDS): k_resblock (SEANet residual block in one launch) - Mimi GPU tests (full-size golden: codes bit-exact), same-box
# A/B against the two-launch path (Mimi only, 32 and 8 sessions), default line + kernel trace + sites
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_mimi_gpu.py tests/test_loaders_gpu.py -m gpu -q --timeout=600 > $O/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_subset.log
run() { # file, label, bench args..., env via VARS
  local file=$1 label=$2; shift 2
  env $VARS timeout 200 python bench.py "$@" --no-cpu-baseline > $O/ab_tmp.log 2>&1
  echo "$label $(grep '"metric"' $O/ab_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))")" >> $O/$file
}
for rep in 1 2; do
  VARS="MMI_MIMI_NO_RES_FUSION=1" run ab_resblock.txt "mimi only B=32 two launches per block" --workload mimi
  VARS="MMI_DUMMY=1" run ab_resblock.txt "mimi only B=32 k_resblock" --workload mimi
done
VARS="MMI_MIMI_NO_RES_FUSION=1" run ab_resblock.txt "mimi only B=8 two launches per block" --workload mimi --batch 8
VARS="MMI_DUMMY=1" run ab_resblock.txt "mimi only B=8 k_resblock" --workload mimi --batch 8
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_default -o duplex -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --launch-lists $O/ll > $O/rocprof_default.log 2>&1
cd $GRAFT_REPO_ROOT
HDR="rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (the default benchmark command: duplex, 32 sessions, 60 steps + 12 warm-up + 248 stagger steps)"
python scripts/rocpd_stats.py /tmp/prof_default/duplex_results.db --header "$HDR" > $O/r02_duplex_b32_k_kernel_stats.csv
python scripts/rocpd_sites.py /tmp/prof_default/duplex_results.db $O/ll --header "$HDR" > $O/r02_duplex_b32_k_sites.csv 2> $O/sites_err.log
tail -5 $O/pytest_gpu_subset.log; cat $O/ab_resblock.txt; grep '"metric"' $O/bench_default.log | cut -c1-300; grep -E "TOTAL|^mimi" $O/r02_duplex_b32_k_sites.csv
