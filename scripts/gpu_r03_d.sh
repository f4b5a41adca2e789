# round 3, call d: the codec gated behind the LM's depth-transformer phase (MMI_DUPLEX_GATE) x stream priorities, same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f frames/s %.0f dom %.1f us' % (d['ms_per_step'], d['p50_ms_per_step'], d['value'], 1e3*d['roofline']['avg_launch_ms']))"; }
timeout 600 python -m pytest tests/test_duplex_gpu.py -q > $O/d_duplex_tests.log 2>&1; echo "duplex tests (gated) rc=$?"; tail -3 $O/d_duplex_tests.log
MMI_DUPLEX_GATE=0 timeout 600 python -m pytest tests/test_duplex_gpu.py -q > $O/d_duplex_tests_ungated.log 2>&1; echo "duplex tests (ungated) rc=$?"; tail -3 $O/d_duplex_tests_ungated.log
for cfg in "1 mimi" "0 mimi" "1 lm" "0 lm" "1 mimi" "serial"; do
  set -- $cfg
  if [ "$1" = "serial" ]; then
    timeout 300 python bench.py --no-cpu-baseline --serial > $O/d_bench.log 2>&1
  else
    MMI_DUPLEX_GATE=$1 MMI_DUPLEX_PRIO=$2 timeout 300 python bench.py --no-cpu-baseline > $O/d_bench.log 2>&1
  fi
  echo "gate/prio = $cfg: $(line $O/d_bench.log)" | tee -a $O/d_summary.txt
done
