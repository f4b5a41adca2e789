# Round 4, GPU call W: after call V's result (no ring split from 4 sessions on; short-ring threshold 1200 rows at two): parity of the
# attention paths, and the default plan at 8 sessions / full ring and 2 sessions / 1800 rows.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
rm -f $O/w_summary.txt
timeout 100 python -m pytest tests/test_lm_gpu.py tests/test_duplex_gpu.py -m gpu -x -q -k "long_ring or tiny_matches or program_switch" > $O/w_pytest.log 2>&1; echo "pytest subset: exit $? $(tail -1 $O/w_pytest.log)" | tee -a $O/w_summary.txt
for cfg in "8 3000" "2 1800" "16 3000"; do
  set -- $cfg
  timeout 60 python bench.py --no-cpu-baseline --no-extras --workload lm --batch $1 --kv-seek $2 --steps 30 --warmup 6 > $O/w_b$1_$2.log 2>&1
  echo "lm B=$1 depth $2, default plan: $(grep '"metric"' $O/w_b$1_$2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))")" | tee -a $O/w_summary.txt
done
