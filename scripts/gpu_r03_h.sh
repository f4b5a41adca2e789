# round 3, call h: which codec half is held behind the LM's phase (MMI_DUPLEX_GATE bit 0 encoder, bit 1 decoder) x priorities
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f frames/s %.0f dom %.1f us' % (d['ms_per_step'], d['p50_ms_per_step'], d['value'], 1e3*d['roofline']['avg_launch_ms']))"; }
timeout 300 python -m pytest tests/test_duplex_gpu.py -q > $O/h_duplex_tests.log 2>&1; echo "duplex tests rc=$?"; tail -3 $O/h_duplex_tests.log
for cfg in "3 mimi" "2 mimi" "1 mimi" "3 tri" "2 tri" "1 tri" "3 mimi"; do
  set -- $cfg
  MMI_DUPLEX_GATE=$1 MMI_DUPLEX_PRIO=$2 timeout 300 python bench.py --no-cpu-baseline > $O/h_bench.log 2>&1
  echo "gate/prio = $cfg: $(line $O/h_bench.log)" | tee -a $O/h_summary.txt
done
for cfg in "3 tri" "2 mimi"; do
  set -- $cfg
  echo "=== MMI_DUPLEX_GATE=$1 MMI_DUPLEX_PRIO=$2" | tee -a $O/h_timeline.txt
  MMI_DUPLEX_GATE=$1 MMI_DUPLEX_PRIO=$2 timeout 300 python scripts/duplex_timeline.py 2>&1 | grep "last of\|isolated frame 3" | tee -a $O/h_timeline.txt
done
