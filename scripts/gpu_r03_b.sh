# round 3, call b: what cross-stream dependencies cost on this stack (scripts/stream_probe.hip), the duplex tests after the
# flow-control fix, and a kernel trace of the pipelined step with queue ids
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w scripts/stream_probe.hip -o /tmp/stream_probe && timeout 120 /tmp/stream_probe > $O/b_stream_probe.txt 2>&1
cat $O/b_stream_probe.txt
GPU_MAX_HW_QUEUES=8 timeout 120 /tmp/stream_probe > $O/b_stream_probe_hwq8.txt 2>&1
grep "^ 2\|^ 3c\|^ 4" $O/b_stream_probe_hwq8.txt | sed 's/^/hwq8: /'
timeout 900 python -m pytest tests/test_duplex_gpu.py -q > $O/b_duplex_tests.log 2>&1; echo "duplex tests rc=$?"
tail -5 $O/b_duplex_tests.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_pipe -o pipe -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 > $O/b_rocprof_pipe.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_overlap.py /tmp/prof_pipe/pipe_results.db 40 > $O/b_pipe_overlap.csv 2>&1
grep "^#" $O/b_pipe_overlap.csv
grep '"metric"' $O/b_rocprof_pipe.log | cut -c1-200
