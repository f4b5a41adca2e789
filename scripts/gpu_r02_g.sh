# round 2, seventh GPU call: same-box A/B of (a) the value-row prefetch in the temporal attention (engine of commit 82f498a
# against the current one, both with whole-tile sharing) and (b) the octet-granular tile sharing of k_gemm_xlds; GPU suite; trace
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 200 python bench.py --workload lm --no-cpu-baseline > $O/ab_$label.log 2>&1
  echo "lm only $label $(grep '"metric"' $O/ab_$label.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))")" >> $O/ab_attn_octets.txt
}
for rep in 1 2; do
  run old_82f498a MMI_LIB_PATH=$GRAFT_REPO_ROOT/ab_old/libmoshi_mi_82f498a.so
  run new_whole_tiles MMI_XLDS_WHOLE_TILES=1
  run new_octets MMI_DUMMY=1
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_default -o duplex -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --launch-lists $O/ll > $O/rocprof_default.log 2>&1
cd $GRAFT_REPO_ROOT
HDR="rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (the default benchmark command: duplex, 32 sessions, 60 steps + 12 warm-up + 248 stagger steps)"
python scripts/rocpd_stats.py /tmp/prof_default/duplex_results.db --header "$HDR" > $O/r02_duplex_b32_g_kernel_stats.csv
python scripts/rocpd_sites.py /tmp/prof_default/duplex_results.db $O/ll --header "$HDR" > $O/r02_duplex_b32_g_sites.csv 2> $O/sites_err.log
grep -E "passed|failed" $O/pytest_gpu.log | tail -3; cat $O/ab_attn_octets.txt; grep -E "^lm,L\.|TOTAL" $O/r02_duplex_b32_g_sites.csv
