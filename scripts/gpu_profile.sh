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
# HBM traffic of the dominant kernel INSIDE the step (the product library, eager launches so that every dispatch is a packet the
# counters can be attributed to).  rocprofv3 --pmc crashed inside the profiler when attached to the Python process in rounds 1-2
# (profiles/r01_logs); tried again each round, bounded, and the standalone-launcher figure stays the committed one if it still does.
for ctr in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && MMI_NO_GRAPH=1 timeout 150 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/prof_pmc_$ctr -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload lm --steps 2 --warmup 1 > $O/rocprof_pmc_$ctr.log 2>&1; rc=$?; echo "pmc $ctr rc=$rc"
  cd $GRAFT_REPO_ROOT
  tail -3 $O/rocprof_pmc_$ctr.log | cut -c1-200
  if [ $rc != 0 ] || ! ls /tmp/prof_pmc_$ctr/*.db > /dev/null 2>&1; then break; fi
  python scripts/rocpd_pmc.py /tmp/prof_pmc_$ctr/pmc_results.db --header "rocprofv3 --pmc $ctr --kernel-trace -- MMI_NO_GRAPH=1 python bench.py --no-cpu-baseline --workload lm --steps 2 --warmup 1 (the LM step of the product library, 32 sessions)" 2>&1 | head -12 > $O/pmc_${ctr}_in_step.csv
done
