# round 3, call a: the new parity cases (reference at full depth, C5's shape, duplex pipeline) + pipelined vs serial bench on ONE box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_duplex_gpu.py -x -q > $O/a_duplex_tests.log 2>&1; echo "duplex tests rc=$?" | tee -a $O/a_summary.txt
tail -5 $O/a_duplex_tests.log
timeout 1200 python -m pytest tests/test_lm_gpu.py -x -q -s -k "benchmark_model or c5_shape or greedy_schedule or 7b_layer_shapes" > $O/a_lm_tests.log 2>&1; echo "lm tests rc=$?" | tee -a $O/a_summary.txt
grep "parity\]" $O/a_lm_tests.log | head -60; tail -5 $O/a_lm_tests.log
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f p95 %.3f frames/s %.0f step_frac %.3f dom %.1f us' % (d['ms_per_step'], d['p50_ms_per_step'], d['p95_ms_per_step'], d['value'], d['roofline']['step']['frac'], 1e3*d['roofline']['avg_launch_ms']))"; }
for v in pipe serial pipe_noprio pipe serial; do
  case $v in
    pipe) E=""; A="";;
    serial) E=""; A="--serial";;
    pipe_noprio) E="MMI_DUPLEX_PRIO=0"; A="";;
  esac
  env $E timeout 300 python bench.py --no-cpu-baseline $A > $O/a_bench_$v.log 2>&1
  echo "duplex b32 $v: $(line $O/a_bench_$v.log)" | tee -a $O/a_summary.txt
done
timeout 300 python bench.py --no-cpu-baseline --steps 200 > $O/a_bench_pipe_200.log 2>&1
echo "duplex b32 pipe 200 steps: $(line $O/a_bench_pipe_200.log)" | tee -a $O/a_summary.txt
