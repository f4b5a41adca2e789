# round 3, call i: host cost of the engine calls and of a pipelined submit
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python scripts/duplex_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/i_timeline.txt
