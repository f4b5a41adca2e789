# round 3, call g: flag hand-offs instead of pending event waits: duplex tests + timeline + bench under gate / events / priority settings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f frames/s %.0f dom %.1f us' % (d['ms_per_step'], d['p50_ms_per_step'], d['value'], 1e3*d['roofline']['avg_launch_ms']))"; }
timeout 300 python -m pytest tests/test_duplex_gpu.py -q > $O/g_duplex_tests.log 2>&1; echo "duplex tests (gated, flags) rc=$?"; tail -3 $O/g_duplex_tests.log
MMI_DUPLEX_GATE=0 timeout 300 python -m pytest tests/test_duplex_gpu.py -q > $O/g_duplex_tests_ungated.log 2>&1; echo "duplex tests (ungated, flags) rc=$?"; tail -3 $O/g_duplex_tests_ungated.log
for cfg in "1 mimi 0" "0 mimi 0"; do
  set -- $cfg
  echo "=== MMI_DUPLEX_GATE=$1 MMI_DUPLEX_PRIO=$2 MMI_DUPLEX_EVENTS=$3" | tee -a $O/g_timeline.txt
  MMI_DUPLEX_GATE=$1 MMI_DUPLEX_PRIO=$2 MMI_DUPLEX_EVENTS=$3 timeout 300 python scripts/duplex_timeline.py 2>&1 | grep -v amdgpu.ids | tee -a $O/g_timeline.txt
done
for cfg in "1 mimi 0" "0 mimi 0" "1 lm 0" "1 0 0" "1 mimi 1" "1 mimi 0" "serial"; do
  set -- $cfg
  if [ "$1" = "serial" ]; then
    timeout 300 python bench.py --no-cpu-baseline --serial > $O/g_bench.log 2>&1
  else
    MMI_DUPLEX_GATE=$1 MMI_DUPLEX_PRIO=$2 MMI_DUPLEX_EVENTS=$3 timeout 300 python bench.py --no-cpu-baseline > $O/g_bench.log 2>&1
  fi
  echo "gate/prio/events = $cfg: $(line $O/g_bench.log)" | tee -a $O/g_summary.txt
done
