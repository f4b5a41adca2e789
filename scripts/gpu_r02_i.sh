# round 2, call i: (a) octet sharing of the depth transformer's N = 1024 GEMMs (k_gemm_xp osplit) off / 2 / default (4), LM only;
# (b) k_conv_wide with incremental gather offsets against the engine of commit d970fc5 (offset table), Mimi only; parity subset;
# default line + kernel trace + sites
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -x -k "full_size or full_width_layers or 7b_layer_shapes or tiny_matches_oracle or two_batch_tiles or round_trip" > $O/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_subset.log
run() { # file, label, bench args..., env via VARS
  local file=$1 label=$2; shift 2
  env $VARS timeout 200 python bench.py "$@" --no-cpu-baseline > $O/ab_tmp.log 2>&1
  echo "$label $(grep '"metric"' $O/ab_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))")" >> $O/$file
}
for rep in 1 2; do
  VARS="MMI_GEMM_OSPLIT=0" run ab_osplit.txt "lm only osplit=0" --workload lm
  VARS="MMI_GEMM_OSPLIT=2" run ab_osplit.txt "lm only osplit=2" --workload lm
  VARS="MMI_DUMMY=1" run ab_osplit.txt "lm only osplit=default(4)" --workload lm
  VARS="MMI_LIB_PATH=$GRAFT_REPO_ROOT/ab_old/libmoshi_mi_d970fc5.so" run ab_conv_offsets.txt "mimi only B=32 d970fc5 (offset table)" --workload mimi
  VARS="MMI_DUMMY=1" run ab_conv_offsets.txt "mimi only B=32 incremental offsets" --workload mimi
done
VARS="MMI_DUMMY=1" run ab_osplit.txt "lm only B=1 default" --workload lm --batch 1
VARS="MMI_GEMM_OSPLIT=0" run ab_osplit.txt "lm only B=1 osplit=0" --workload lm --batch 1
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_default -o duplex -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --launch-lists $O/ll > $O/rocprof_default.log 2>&1
cd $GRAFT_REPO_ROOT
HDR="rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (the default benchmark command: duplex, 32 sessions, 60 steps + 12 warm-up + 248 stagger steps)"
python scripts/rocpd_stats.py /tmp/prof_default/duplex_results.db --header "$HDR" > $O/r02_duplex_b32_i_kernel_stats.csv
python scripts/rocpd_sites.py /tmp/prof_default/duplex_results.db $O/ll --header "$HDR" > $O/r02_duplex_b32_i_sites.csv 2> $O/sites_err.log
tail -3 $O/pytest_gpu_subset.log; cat $O/ab_osplit.txt $O/ab_conv_offsets.txt; grep '"metric"' $O/bench_default.log | cut -c1-300; grep -E "^lm,dep|^lm,L.attn|TOTAL|^mimi" $O/r02_duplex_b32_i_sites.csv
