# Round 6, call A: same-box A/B of the depth transformer's own 16-row tile (MMI_DEP_TILE) and the one-round-trip GEMM
# (MMI_GEMM_ONCE) on the 32-session duplex step, serial schedule (LM + codec back to back) and pipelined, plus the per-site
# rocprofv3 table of the new default and of round 5's plan.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f frames/s %.0f' % (d['ms_per_step'], d['p50_ms_per_step'], d['value']))"; }
run() {   # name, env..., -- bench args
  name=$1; shift
  ( env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extras $BARGS ) > $O/ab_$name.log 2>&1
  echo "$name: $(line $O/ab_$name.log)" | tee -a $O/ab_lines.txt
}
rm -f $O/ab_lines.txt
# parity first: the new default path against the oracle / the reference's goldens at the sizes that take it
timeout 900 python -m pytest tests/test_b_lm_gpu.py -x -q -k "tiny_matches or full_width_layers or full_depth_32 or lds_resident or reproducible_between or 7b_layer or benchmark_kernels" > $O/pytest_ab.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_ab.log | cut -c1-200
BARGS="--serial"
run serial_old MMI_DEP_TILE=32 MMI_GEMM_ONCE=0
run serial_once MMI_DEP_TILE=32
run serial_tile MMI_GEMM_ONCE=0
run serial_new X=1
BARGS=""
run pipe_old MMI_DEP_TILE=32 MMI_GEMM_ONCE=0
run pipe_new X=1
run pipe_old2 MMI_DEP_TILE=32 MMI_GEMM_ONCE=0
run pipe_new2 X=1
BARGS="--workload lm --batch 1"
run c3_old MMI_GEMM_ONCE=0
run c3_new X=1
# per-site kernel time, serial schedule, new default and round 5's plan
prof() {
  name=$1; shift
  cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o duplex -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --serial --launch-lists $O/launch_lists_$name > $O/rocprof_$name.log 2>&1
  cd $GRAFT_REPO_ROOT
  python scripts/rocpd_stats.py /tmp/prof_$name/duplex_results.db --header "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras --serial ($name)" > $O/duplex_b32_serial_${name}_kernel_stats.csv
  python scripts/rocpd_sites.py /tmp/prof_$name/duplex_results.db $O/launch_lists_$name --header "per-site kernel time, serial schedule, 32 sessions ($name)" > $O/duplex_b32_serial_${name}_sites.csv
  grep "^lm" $O/duplex_b32_serial_${name}_sites.csv | cut -c1-120
}
prof new X=1
prof old MMI_DEP_TILE=32 MMI_GEMM_ONCE=0
cat $O/ab_lines.txt
