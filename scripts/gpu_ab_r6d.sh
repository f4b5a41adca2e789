# (needs scripts/experiments/mimi_tr_grouped.patch applied: MMI_MIMI_TR is read by the patched engine only)
# round 6, second session: the grouped Mimi transformer (mimi_tr_kernels.h) - parity first, then same-box A/B of the three forms
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
rm -f $O/t_*.log $O/t_lines.txt
( timeout 900 python -m pytest tests/test_a_mimi_gpu.py -x -q ) > $O/t_pytest_mimi.log 2>&1; echo "pytest mimi rc=$?" | tee -a $O/t_lines.txt
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f value %.0f' % (d['ms_per_step'], d['p50_ms_per_step'], d['value']))"; }
run() { name=$1; shift; ( env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extras $BARGS ) > $O/t_$name.log 2>&1; echo "$name: $(line $O/t_$name.log)" | tee -a $O/t_lines.txt; }
for mode in launches phases persist; do
  BARGS="--workload mimi --batch 8"; run mimi_b8_$mode MMI_MIMI_TR=$mode
  BARGS="--workload mimi --batch 32"; run mimi_b32_$mode MMI_MIMI_TR=$mode
done
for mode in launches persist launches persist; do
  BARGS=""; run duplex_b32_$mode MMI_MIMI_TR=$mode
done
BARGS="--serial"; run duplex_serial_launches MMI_MIMI_TR=launches
BARGS="--serial"; run duplex_serial_persist MMI_MIMI_TR=persist
cat $O/t_lines.txt
