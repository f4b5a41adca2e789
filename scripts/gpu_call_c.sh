# Round-5 GPU call C: the round's check as the driver runs it, the profile of the default configuration, the SQ passes.
cd $GRAFT_REPO_ROOT
bash scripts/gpu_check.sh
O=$GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_serial -o duplex -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --serial --launch-lists $O/launch_lists > $O/rocprof_serial.log 2>&1 ); echo "rocprof serial rc=$?"
python scripts/rocpd_stats.py /tmp/prof_serial/duplex_results.db --header "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras --serial   (duplex, 32 sessions, one stream)" > $O/r05_duplex_b32_serial_kernel_stats.csv
python scripts/rocpd_sites.py /tmp/prof_serial/duplex_results.db $O/launch_lists --last 100 --header "per-site kernel time, serial schedule, 32 sessions, mid-run ring depth" > $O/r05_duplex_b32_serial_sites.csv
grep "L\.\|TOTAL\|dep\.\|text" $O/r05_duplex_b32_serial_sites.csv | cut -c1-160 | head -30
bash scripts/gpu_sq_q8.sh
