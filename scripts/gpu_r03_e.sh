# round 3, call e: per-phase timeline of the duplex pipeline under the priority / gate settings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
for cfg in "0 mimi" "0 0" "1 mimi" "0 lm"; do
  set -- $cfg
  echo "=== MMI_DUPLEX_GATE=$1 MMI_DUPLEX_PRIO=$2" | tee -a $O/e_timeline.txt
  MMI_DUPLEX_GATE=$1 MMI_DUPLEX_PRIO=$2 timeout 300 python scripts/duplex_timeline.py 2>&1 | grep -v amdgpu.ids | tee -a $O/e_timeline.txt
done
echo "=== GPU_MAX_HW_QUEUES=8 MMI_DUPLEX_GATE=0 MMI_DUPLEX_PRIO=0" | tee -a $O/e_timeline.txt
GPU_MAX_HW_QUEUES=8 MMI_DUPLEX_GATE=0 MMI_DUPLEX_PRIO=0 timeout 300 python scripts/duplex_timeline.py 2>&1 | grep -v amdgpu.ids | tee -a $O/e_timeline.txt
