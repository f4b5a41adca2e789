# round 2, call n: octet sharing of the fused-norm GEMMs (A/B, LM only); per-site tables at 64 sessions (bf16 and int8 linears): where the
# step from 32 to 64 sessions goes; default line of the tree
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
run() { # file, label, bench args..., env via VARS
  local file=$1 label=$2; shift 2
  env $VARS timeout 200 python bench.py "$@" --no-cpu-baseline > $O/ab_tmp.log 2>&1
  echo "$label $(grep '"metric"' $O/ab_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))")" >> $O/$file
}
for rep in 1 2; do
  VARS="MMI_GEMM_OSPLIT_NORM=0" run ab_osplit_norm.txt "lm only fused-norm GEMMs whole tiles" --workload lm
  VARS="MMI_DUMMY=1" run ab_osplit_norm.txt "lm only fused-norm GEMMs shared in octets" --workload lm
done
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.log 2>&1
for q in none q8; do
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$q -o duplex -- python $GRAFT_REPO_ROOT/bench.py --batch 64 --quant $q --no-cpu-baseline --launch-lists $O/ll_$q > $O/rocprof_b64_$q.log 2>&1
  cd $GRAFT_REPO_ROOT
  HDR="rocprofv3 --kernel-trace --stats -- python bench.py --batch 64 --quant $q --no-cpu-baseline"
  python scripts/rocpd_stats.py /tmp/prof_$q/duplex_results.db --header "$HDR" > $O/r02_duplex_b64_${q}_n_kernel_stats.csv
  python scripts/rocpd_sites.py /tmp/prof_$q/duplex_results.db $O/ll_$q --header "$HDR" > $O/r02_duplex_b64_${q}_n_sites.csv 2>> $O/sites_err.log
  grep '"metric"' $O/rocprof_b64_$q.log | cut -c1-260
done
cat $O/ab_osplit_norm.txt; grep '"metric"' $O/bench_default.log | cut -c1-300; grep -E "^lm|TOTAL" $O/r02_duplex_b64_none_n_sites.csv; grep -E "^lm|TOTAL" $O/r02_duplex_b64_q8_n_sites.csv
