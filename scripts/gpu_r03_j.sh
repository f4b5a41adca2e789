# round 3, call j: relaxed polling in the hand-off flags (no L2 invalidation per poll): timeline + bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f frames/s %.0f dom %.1f us' % (d['ms_per_step'], d['p50_ms_per_step'], d['value'], 1e3*d['roofline']['avg_launch_ms']))"; }
timeout 300 python scripts/duplex_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/j_timeline.txt
for cfg in "3 mimi" "1 mimi" "2 mimi" "3 0" "serial" "3 mimi"; do
  set -- $cfg
  if [ "$1" = "serial" ]; then
    timeout 300 python bench.py --no-cpu-baseline --serial > $O/j_bench.log 2>&1
  else
    MMI_DUPLEX_GATE=$1 MMI_DUPLEX_PRIO=$2 timeout 300 python bench.py --no-cpu-baseline > $O/j_bench.log 2>&1
  fi
  echo "gate/prio = $cfg: $(line $O/j_bench.log)" | tee -a $O/j_summary.txt
done
