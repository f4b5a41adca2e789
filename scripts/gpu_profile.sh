# rocprofv3 evidence for profiles/ (gpurun -- 'bash scripts/gpu_profile.sh'): kernel stats + per-site table of the default
# benchmark command in its SERIAL schedule (the tracer serialises the LM's queue against the codec's, so the pipelined overlap
# cannot be seen in a kernel trace - the pipeline's own timeline is in the bench line and scripts/duplex_timeline.py), the queue
# view of the pipelined run, and the stream probe.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_serial -o duplex -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --serial --launch-lists $O/launch_lists > $O/rocprof_serial.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py /tmp/prof_serial/duplex_results.db --header "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --serial   (duplex, 32 sessions, one stream)" > $O/duplex_b32_serial_kernel_stats.csv
python scripts/rocpd_sites.py /tmp/prof_serial/duplex_results.db $O/launch_lists --header "per-site kernel time, serial schedule, 32 sessions" > $O/duplex_b32_serial_sites.csv
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_pipe -o duplex -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --launch-lists $O/launch_lists_pipe > $O/rocprof_pipe.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py /tmp/prof_pipe/duplex_results.db --header "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (the default benchmark command: duplex, 32 sessions, three streams)" > $O/duplex_b32_pipelined_kernel_stats.csv
python scripts/rocpd_sites.py /tmp/prof_pipe/duplex_results.db $O/launch_lists_pipe --header "per-site kernel time, pipelined schedule (under the tracer the LM's queue does not overlap the codec's), 32 sessions" > $O/duplex_b32_pipelined_sites.csv
python scripts/rocpd_overlap.py /tmp/prof_pipe/duplex_results.db 40 | grep "^#" > $O/duplex_b32_pipelined_queues.txt
timeout 300 python scripts/duplex_timeline.py 2>&1 | grep -v amdgpu.ids > $O/duplex_timeline.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w scripts/stream_probe.hip -o /tmp/stream_probe && timeout 120 /tmp/stream_probe > $O/stream_probe.txt 2>&1
head -12 $O/duplex_b32_serial_kernel_stats.csv | cut -c1-150; grep "L\.\|TOTAL" $O/duplex_b32_serial_sites.csv | head -12; cat $O/duplex_b32_pipelined_queues.txt; grep '"metric"' $O/rocprof_pipe.log | cut -c1-200
bash scripts/gpu_pmc.sh
