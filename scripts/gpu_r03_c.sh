# round 3, call c: does the codec overlap the LM at all?  kernel traces of the pipelined step under the three priority settings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f frames/s %.0f' % (d['ms_per_step'], d['p50_ms_per_step'], d['value']))"; }
for v in 0 mimi lm; do
  cd /tmp && MMI_DUPLEX_PRIO=$v timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_pipe_$v -o pipe -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 > $O/c_rocprof_pipe_$v.log 2>&1
  cd $GRAFT_REPO_ROOT
  python scripts/rocpd_overlap.py /tmp/prof_pipe_$v/pipe_results.db 40 > $O/c_pipe_overlap_$v.csv 2>&1
  echo "prio=$v (traced): $(line $O/c_rocprof_pipe_$v.log)"; grep "^# queue\|^# window" $O/c_pipe_overlap_$v.csv
  MMI_DUPLEX_PRIO=$v timeout 300 python bench.py --no-cpu-baseline > $O/c_bench_$v.log 2>&1
  echo "prio=$v (untraced): $(line $O/c_bench_$v.log)"
done
timeout 300 python bench.py --no-cpu-baseline --serial > $O/c_bench_serial.log 2>&1
echo "serial (untraced): $(line $O/c_bench_serial.log)"
