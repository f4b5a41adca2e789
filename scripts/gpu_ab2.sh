set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
MMI_GEMM_NTW=1 timeout 300 python bench.py --no-cpu-baseline > $O/ab2_ntw1_a.log 2>&1
timeout 300 python bench.py --no-cpu-baseline > $O/ab2_plan_a.log 2>&1
MMI_GEMM_NTW=1 timeout 300 python bench.py --no-cpu-baseline > $O/ab2_ntw1_b.log 2>&1
timeout 300 python bench.py --no-cpu-baseline > $O/ab2_plan_b.log 2>&1
timeout 600 python -m pytest tests/test_lm_gpu.py -m gpu -q --timeout=600 -k "full_width or rows_independent or golden" > $O/ab2_pytest.log 2>&1
for f in ab2_ntw1_a ab2_plan_a ab2_ntw1_b ab2_plan_b; do echo $f $(grep '"metric"' $O/$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f frames/s %.0f' % (d['ms_per_step'], d['p50_ms_per_step'], d['value']))"); done
tail -3 $O/ab2_pytest.log
