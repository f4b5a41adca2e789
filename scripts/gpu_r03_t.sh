# round 3, call t: the gated linear_in shared out in row octets (MMI_GATE_OCT=0: whole tiles): parity subset + A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f frames/s %.0f dom %.1f us' % (d['ms_per_step'], d['p50_ms_per_step'], d['value'], 1e3*d['roofline']['avg_launch_ms']))"; }
timeout 900 python -m pytest tests/test_lm_gpu.py -q -k "benchmark_kernels or two_batch_tiles or lds_resident" > $O/t_lm_tests.log 2>&1; echo "lm tests rc=$?"; tail -3 $O/t_lm_tests.log
for v in oct tiles oct tiles; do
  E=""; [ $v = tiles ] && E="MMI_GATE_OCT=0"
  env $E timeout 300 python bench.py --no-cpu-baseline --workload lm > $O/t_bench.log 2>&1
  echo "lm only $v: $(line $O/t_bench.log)" | tee -a $O/t_summary.txt
done
for v in oct tiles; do
  E=""; [ $v = tiles ] && E="MMI_GATE_OCT=0"
  env $E timeout 300 python bench.py --no-cpu-baseline > $O/t_bench.log 2>&1
  echo "duplex $v: $(line $O/t_bench.log)" | tee -a $O/t_summary.txt
  env $E timeout 300 python bench.py --no-cpu-baseline --workload lm --batch 64 > $O/t_bench.log 2>&1
  echo "lm only b64 $v: $(line $O/t_bench.log)" | tee -a $O/t_summary.txt
done
