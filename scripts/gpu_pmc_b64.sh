# L2 request counters over the bf16 LM step at 64 and at 32 sessions (eager launches, product library): what the second batch tile costs
# k_gemm_xlds in L2 requests.  gpurun -- 'bash scripts/gpu_pmc_b64.sh'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
TCC="TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum"
for B in 64 32; do
  cd /tmp && MMI_NO_GRAPH=1 timeout 200 rocprofv3 --pmc $TCC --kernel-trace -d /tmp/b${B}_TCC -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --workload lm --batch $B --steps 2 --warmup 1 > $O/pmc_b${B}_TCC.log 2>&1; echo "B=$B rc=$?"
  cd $GRAFT_REPO_ROOT
  PMC_ROWS=40 python scripts/rocpd_pmc.py /tmp/b${B}_TCC/pmc_results.db --header "rocprofv3 --pmc $TCC --kernel-trace -- MMI_NO_GRAPH=1 python bench.py --workload lm --batch $B --steps 2 (bf16; raw counter values in the avg_KiB column)" --clusters k_gemm_x --by-duration 2>&1 | grep "k_gemm_x\|^#" > $O/r04_pmc_tcc_lm_b${B}.csv
done
cut -c1-200 $O/r04_pmc_tcc_lm_b64.csv | head -40
cut -c1-200 $O/r04_pmc_tcc_lm_b32.csv | head -40
