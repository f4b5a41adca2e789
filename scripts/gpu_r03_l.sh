# round 3, call l: the gate kept by the host (no polling waves / pending waits through the LM's temporal phase)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f frames/s %.0f dom %.1f us' % (d['ms_per_step'], d['p50_ms_per_step'], d['value'], 1e3*d['roofline']['avg_launch_ms']))"; }
timeout 300 python -m pytest tests/test_duplex_gpu.py -q > $O/l_duplex_tests.log 2>&1; echo "duplex tests rc=$?"; tail -3 $O/l_duplex_tests.log
for cfg in "1 mimi" "0 mimi" "1 0" "1 mimi"; do
  set -- $cfg
  MMI_DUPLEX_HOSTGATE=$1 MMI_DUPLEX_PRIO=$2 timeout 300 python bench.py --no-cpu-baseline > $O/l_bench.log 2>&1
  echo "hostgate/prio = $cfg: $(line $O/l_bench.log)" | tee -a $O/l_summary.txt
done
timeout 300 python scripts/duplex_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/l_timeline.txt
