# round 3, call k: distinct hardware queues without priorities (GPU_MAX_HW_QUEUES=8) vs the priority pools
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f frames/s %.0f dom %.1f us' % (d['ms_per_step'], d['p50_ms_per_step'], d['value'], 1e3*d['roofline']['avg_launch_ms']))"; }
for cfg in "8 3 0" "8 1 0" "8 0 0" "16 3 0" "4 3 lm" "8 3 mimi"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 MMI_DUPLEX_GATE=$2 MMI_DUPLEX_PRIO=$3 timeout 300 python bench.py --no-cpu-baseline > $O/k_bench.log 2>&1
  echo "hwq/gate/prio = $cfg: $(line $O/k_bench.log)" | tee -a $O/k_summary.txt
done
echo "=== GPU_MAX_HW_QUEUES=8 MMI_DUPLEX_PRIO=0" | tee $O/k_timeline.txt
GPU_MAX_HW_QUEUES=8 MMI_DUPLEX_PRIO=0 timeout 300 python scripts/duplex_timeline.py 2>&1 | grep "isolated frame 3\|last of" | tee -a $O/k_timeline.txt
