cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
line() { python -c "import sys,json; d=json.loads([l for l in open('$1') if '\"metric\"' in l][-1]); print('ms/step %.3f p50 %.3f frames/s %.0f' % (d['ms_per_step'], d['p50_ms_per_step'], d['value']))"; }
timeout 900 python -m pytest tests/test_mimi_gpu.py tests/test_duplex_gpu.py -m gpu -q -x --timeout=600 > $O/x_mimi_tests.log 2>&1; echo "mimi+duplex gpu tests rc=$?"; tail -3 $O/x_mimi_tests.log
rm -f $O/x_summary.txt
for rep in 1 2; do for nf in 1 0; do
  if [ $nf = 1 ]; then export MMI_MIMI_NO_FIN_FUSION=1; else unset MMI_MIMI_NO_FIN_FUSION; fi
  timeout 200 python bench.py --no-cpu-baseline --workload mimi > $O/x_b.log 2>&1; echo "mimi b32 absorb=$((1-nf)): $(line $O/x_b.log)" | tee -a $O/x_summary.txt
  timeout 200 python bench.py --no-cpu-baseline > $O/x_b.log 2>&1; echo "duplex b32 absorb=$((1-nf)): $(line $O/x_b.log)" | tee -a $O/x_summary.txt
  timeout 200 python bench.py --no-cpu-baseline --serial > $O/x_b.log 2>&1; echo "duplex b32 serial absorb=$((1-nf)): $(line $O/x_b.log)" | tee -a $O/x_summary.txt
done; done
