# Round 6, call C: parity of the new default path (sampler, once), then per-site table.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python -m pytest tests/test_b_lm_gpu.py -x -q -k "sampler or sampling or sampled or tiny_matches or full_width_layers or lds_resident or reproducible_between or 7b_layer or benchmark_kernels or rng" > $O/pytest_c.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_c.log | cut -c1-200
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f frames/s %.0f' % (d['ms_per_step'], d['p50_ms_per_step'], d['value']))"; }
( timeout 300 python bench.py --no-cpu-baseline --no-extras ) > $O/c_pipe.log 2>&1; echo "pipe: $(line $O/c_pipe.log)"
( timeout 300 python bench.py --no-cpu-baseline --no-extras --serial ) > $O/c_serial.log 2>&1; echo "serial: $(line $O/c_serial.log)"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o duplex -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --serial --launch-lists $O/launch_lists_c > $O/rocprof_c.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_sites.py /tmp/prof_c/duplex_results.db $O/launch_lists_c --header "per-site kernel time, serial schedule, 32 sessions (call C)" > $O/duplex_b32_serial_c_sites.csv
grep "^lm" $O/duplex_b32_serial_c_sites.csv | cut -c1-120
