# round 3, call f: why does the LM graph run slower on the pipeline's streams?  graph launches per stream kind (probe), and the
# serial loop on non-default torch streams
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w scripts/stream_probe.hip -o /tmp/stream_probe && timeout 120 /tmp/stream_probe > $O/f_stream_probe.txt 2>&1
grep "1e" $O/f_stream_probe.txt
timeout 300 python scripts/serial_on_stream.py 2>&1 | grep -v amdgpu.ids | tee $O/f_serial_on_stream.txt
