# (needs scripts/experiments/rvq_grouped.patch or rvq_select_last_arriver.patch applied: MMI_RVQ is read by the patched engine only)
# round 6: the residual quantiser as one launch (the select step in the last-arriving workgroup) - parity, then same-box A/B against the level-by-level launch list
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
rm -f $O/r_*.log $O/r_lines.txt
( timeout 900 python -m pytest tests/test_a_mimi_gpu.py tests/test_c_duplex_gpu.py -x -q ) > $O/r_pytest_mimi.log 2>&1; echo "pytest mimi+duplex rc=$?" | tee -a $O/r_lines.txt
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f value %.0f' % (d['ms_per_step'], d['p50_ms_per_step'], d['value']))"; }
run() { name=$1; shift; ( env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extras $BARGS ) > $O/r_$name.log 2>&1; echo "$name: $(line $O/r_$name.log)" | tee -a $O/r_lines.txt; }
for mode in launches fused launches fused launches fused; do
  BARGS="--workload mimi --batch 8"; run mimi_b8_$mode MMI_RVQ=$mode
  BARGS="--workload mimi --batch 32"; run mimi_b32_$mode MMI_RVQ=$mode
done
for mode in launches fused launches fused launches fused; do
  BARGS=""; run duplex_b32_$mode MMI_RVQ=$mode
done
BARGS="--batch 64 --quant q8 --kv fp8"; run c5_launches MMI_RVQ=launches
BARGS="--batch 64 --quant q8 --kv fp8"; run c5_fused MMI_RVQ=fused
cat $O/r_lines.txt
