# SQ counter passes over the int8 x int8 temporal GEMMs at C5's shape, three counters per pass (a pass with
# seven did not finish in 200 s in this round's first attempt).  gpurun -- 'bash scripts/gpu_sq_q8.sh'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
HDR="MMI_NO_GRAPH=1 python bench.py --no-cpu-baseline --no-extras --workload lm --steps 2 --warmup 1 --batch 64"
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS"; do
  i=$((i+1))
  for q in q8 none; do
    ( cd /tmp && MMI_NO_GRAPH=1 timeout 280 rocprofv3 --pmc $ctrs --kernel-trace -d /tmp/sq_${q}_$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --workload lm --steps 2 --warmup 1 --batch 64 --quant $q > $O/sq_${q}_$i.log 2>&1 ); echo "sq $q pass $i rc=$?"
    PMC_ROWS=60 python scripts/rocpd_pmc.py /tmp/sq_${q}_$i/pmc_results.db --header "rocprofv3 --pmc $ctrs -- $HDR --quant $q (raw counter values in the avg_KiB column)" --clusters "k_gemm" --by-duration > $O/r05_pmc_sq_${q}_b64_pass$i.csv 2>&1
  done
done
for f in $O/r05_pmc_sq_*.csv; do echo "== $f"; grep -A30 "clusters" $f | grep "k_gemm_x" | cut -c1-150 | head -14; done
