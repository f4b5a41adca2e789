# round 3, call o: soak - the default benchmark 10 times + the duplex tests 5 times on one box; any fault?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f frames/s %.0f' % (d['ms_per_step'], d['p50_ms_per_step'], d['value']))"; }
for i in 1 2 3 4 5 6 7 8 9 10; do
  MMI_BENCH_TRACE=1 timeout 300 python bench.py --no-cpu-baseline --steps 200 > $O/o_bench_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc: $(line $O/o_bench_$i.log)" | tee -a $O/o_summary.txt
  grep -i "fault\|error" $O/o_bench_$i.log | head -3
done
for i in 1 2 3 4 5; do
  timeout 300 python -m pytest tests/test_duplex_gpu.py -q > $O/o_tests_$i.log 2>&1; echo "tests $i rc=$?" | tee -a $O/o_summary.txt
done
