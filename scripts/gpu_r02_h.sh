# round 2, eighth GPU call (first of the re-created container: call g's output was lost): the octet-granular tile sharing of
# k_gemm_xlds on hardware - full-width parity subset, same-box A/B against whole-tile sharing, default line, kernel trace + sites
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests -m gpu -q --timeout=500 -x -k "lds_resident or full_width_layers or two_batch_tiles or 7b_layer_shapes" > $O/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_subset.log
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 200 python bench.py --workload lm --no-cpu-baseline > $O/ab_$label.log 2>&1
  echo "lm only $label $(grep '"metric"' $O/ab_$label.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))")" >> $O/ab_octets.txt
}
for rep in 1 2; do
  run whole_tiles MMI_XLDS_WHOLE_TILES=1
  run octets MMI_DUMMY=1
done
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_default -o duplex -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --launch-lists $O/ll > $O/rocprof_default.log 2>&1
cd $GRAFT_REPO_ROOT
HDR="rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (the default benchmark command: duplex, 32 sessions, 60 steps + 12 warm-up + 248 stagger steps)"
python scripts/rocpd_stats.py /tmp/prof_default/duplex_results.db --header "$HDR" > $O/r02_duplex_b32_h_kernel_stats.csv
python scripts/rocpd_sites.py /tmp/prof_default/duplex_results.db $O/ll --header "$HDR" > $O/r02_duplex_b32_h_sites.csv 2> $O/sites_err.log
tail -3 $O/pytest_gpu_subset.log; cat $O/ab_octets.txt; grep '"metric"' $O/bench_default.log | cut -c1-400; grep -E "^lm,|TOTAL" $O/r02_duplex_b32_h_sites.csv
