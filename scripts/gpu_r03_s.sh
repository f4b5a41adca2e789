# round 3, call s: where in the LM step the codec is let in (MMI_LM_PHASE_LAYER: start of that temporal layer; unset: depth transformer)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f frames/s %.0f dom %.1f us' % (d['ms_per_step'], d['p50_ms_per_step'], d['value'], 1e3*d['roofline']['avg_launch_ms']))"; }
for v in none 31 29 26 20 none; do
  E=""; [ $v != none ] && E="MMI_LM_PHASE_LAYER=$v"
  env $E timeout 300 python bench.py --no-cpu-baseline > $O/s_bench.log 2>&1
  echo "phase point $v: $(line $O/s_bench.log)" | tee -a $O/s_summary.txt
done
for v in none 29; do
  E=""; [ $v != none ] && E="MMI_LM_PHASE_LAYER=$v"
  env $E timeout 300 python bench.py --no-cpu-baseline --batch 64 --quant q8 > $O/s_bench.log 2>&1
  echo "b64 q8 phase point $v: $(line $O/s_bench.log)" | tee -a $O/s_summary.txt
done
timeout 300 python bench.py --no-cpu-baseline --batch 64 --quant q8 --serial > $O/s_bench.log 2>&1
echo "b64 q8 serial: $(line $O/s_bench.log)" | tee -a $O/s_summary.txt
timeout 300 python bench.py --no-cpu-baseline --batch 64 --quant fp8 > $O/s_bench.log 2>&1
echo "b64 fp8: $(line $O/s_bench.log)" | tee -a $O/s_summary.txt
timeout 300 python bench.py --no-cpu-baseline --batch 64 > $O/s_bench.log 2>&1
echo "b64 bf16: $(line $O/s_bench.log)" | tee -a $O/s_summary.txt
