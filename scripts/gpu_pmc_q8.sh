# Counter passes over the LM step at C5's shape (64 sessions, int8 linears) on the product library, eager launches: what binds the
# int8 GEMMs on k_gemm_xp (default) and on k_gemm_xlds (MMI_GEMM_LDS=1).  gpurun -- 'bash scripts/gpu_pmc_q8.sh'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY"
TCC="TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum"
for var in xp xlds; do
  if [ $var = xlds ]; then export MMI_GEMM_LDS=1; else unset MMI_GEMM_LDS; fi
  for set in SQ TCC; do
    eval ctrs=\$$set
    cd /tmp && MMI_NO_GRAPH=1 timeout 150 rocprofv3 --pmc $ctrs --kernel-trace -d /tmp/q8_${var}_$set -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload lm --batch 64 --quant q8 --steps 2 --warmup 1 > $O/q8_pmc_${var}_$set.log 2>&1; echo "$var $set rc=$?"
    cd $GRAFT_REPO_ROOT
    PMC_ROWS=40 python scripts/rocpd_pmc.py /tmp/q8_${var}_$set/pmc_results.db --header "rocprofv3 --pmc $ctrs -- MMI_NO_GRAPH=1 ${var} python bench.py --workload lm --batch 64 --quant q8 --steps 2 (raw counter values in the avg_KiB column)" --clusters k_gemm_x --by-duration 2>&1 | grep "k_gemm_x\|^#" > $O/q8_pmc_${var}_$set.csv
  done
done
grep -h "clusters" -A12 $O/q8_pmc_xp_SQ.csv | cut -c1-160 | head -30
