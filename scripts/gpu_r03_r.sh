# round 3, call r: RMSNorm folded across out_proj -> linear_in (no norm2 launch): parity subset + A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f frames/s %.0f dom %.1f us' % (d['ms_per_step'], d['p50_ms_per_step'], d['value'], 1e3*d['roofline']['avg_launch_ms']))"; }
timeout 900 python -m pytest tests/test_lm_gpu.py -q -s -k "benchmark_kernels or full_width_layers or two_batch_tiles or lds_resident" > $O/r_lm_tests_fold.log 2>&1; echo "lm tests (fold) rc=$?"; grep -A10 "parity\] golden" $O/r_lm_tests_fold.log | grep "golden\|text\|audio3\|audio7"; tail -3 $O/r_lm_tests_fold.log
MMI_NO_NORM_FOLD=1 timeout 900 python -m pytest tests/test_lm_gpu.py -q -s -k "benchmark_kernels" > $O/r_lm_tests_nofold.log 2>&1; echo "lm tests (nofold) rc=$?"; grep -A10 "parity\] golden" $O/r_lm_tests_nofold.log | grep "golden\|text\|audio3\|audio7"; tail -3 $O/r_lm_tests_nofold.log
for v in fold nofold fold1off fold nofold; do
  E=""; [ $v = nofold ] && E="MMI_NO_NORM_FOLD=1"; [ $v = fold1off ] && E="MMI_NO_NORM_FOLD1=1"
  env $E timeout 300 python bench.py --no-cpu-baseline --workload lm > $O/r_bench.log 2>&1
  echo "lm only $v: $(line $O/r_bench.log)" | tee -a $O/r_summary.txt
done
for v in fold nofold fold; do
  E=""; [ $v = nofold ] && E="MMI_NO_NORM_FOLD=1"
  env $E timeout 300 python bench.py --no-cpu-baseline > $O/r_bench.log 2>&1
  echo "duplex $v: $(line $O/r_bench.log)" | tee -a $O/r_summary.txt
done
for v in fold nofold; do
  E=""; [ $v = nofold ] && E="MMI_NO_NORM_FOLD=1"
  env $E timeout 300 python bench.py --no-cpu-baseline --workload lm --batch 64 > $O/r_bench.log 2>&1
  echo "lm only b64 $v: $(line $O/r_bench.log)" | tee -a $O/r_summary.txt
done
