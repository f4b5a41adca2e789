# round 3, call m: p50 from the pipeline's own timeline (host flush, no device-side waiters) + queue placement lottery (MMI_DUPLEX_PAD)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f p95 %.3f frames/s %.0f dom %.1f us' % (d['ms_per_step'], d['p50_ms_per_step'], d['p95_ms_per_step'], d['value'], 1e3*d['roofline']['avg_launch_ms']))"; }
for cfg in "0 mimi" "1 mimi" "2 mimi" "3 mimi" "0 tri" "1 tri" "2 tri" "0 lm" "1 lm" "0 mimi"; do
  set -- $cfg
  MMI_BENCH_TRACE=1 MMI_DUPLEX_PAD=$1 MMI_DUPLEX_PRIO=$2 timeout 300 python bench.py --no-cpu-baseline > $O/m_bench_$1_$2.log 2>&1; cp $O/m_bench_$1_$2.log $O/m_bench.log
  echo "pad/prio = $cfg: $(line $O/m_bench.log)" | tee -a $O/m_summary.txt
done
