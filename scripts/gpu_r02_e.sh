# round 2, fifth GPU call: Mimi LayerNorm fusion on hardware (GPU suite incl. the reference goldens), same-box A/B, Mimi-only C2 line
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export MMI_MIMI_NO_LN_FUSION=1; else unset MMI_MIMI_NO_LN_FUSION; fi
  timeout 200 python bench.py --workload mimi --batch 32 --no-cpu-baseline > $O/ab_ln_nofuse$v.log 2>&1
  echo "mimi only B=32 MMI_MIMI_NO_LN_FUSION=$v $(grep '"metric"' $O/ab_ln_nofuse$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))")" >> $O/ab_ln.txt
done
unset MMI_MIMI_NO_LN_FUSION
timeout 200 python bench.py --workload mimi --batch 8 --no-cpu-baseline > $O/bench_mimi_b8.log 2>&1
( time timeout 900 python bench.py ) > $O/bench_default.log 2>&1
tail -n 3 $O/smoke.log; grep -E "passed|failed" $O/pytest_gpu.log | tail -3; cat $O/ab_ln.txt; grep '"metric"' $O/bench_mimi_b8.log | cut -c1-200; grep '"metric"' $O/bench_default.log | cut -c1-300
