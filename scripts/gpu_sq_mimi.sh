# SQ counter passes over the codec's kernels at 32 streams (round 6): how busy the matrix core is in the fp32 convolutions and how long
# their waves wait ("12 % of the fp32 MFMA peak").  Three counters per pass.  gpurun -- 'bash scripts/gpu_sq_mimi.sh'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
HDR="MMI_NO_GRAPH=1 python bench.py --no-cpu-baseline --no-extras --workload mimi --steps 6 --warmup 2 --batch 32"
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  ( cd /tmp && MMI_NO_GRAPH=1 timeout 280 rocprofv3 --pmc $ctrs --kernel-trace -d /tmp/sqm_$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --workload mimi --steps 6 --warmup 2 --batch 32 > $O/sqm_$i.log 2>&1 ); echo "sq mimi pass $i rc=$?"
  PMC_ROWS=400 python scripts/rocpd_pmc.py /tmp/sqm_$i/pmc_results.db --header "rocprofv3 --pmc $ctrs -- $HDR (raw counter values in the avg_KiB column)" > $O/r06_pmc_sq_mimi_b32_pass$i.csv 2>&1
  head -3 $O/r06_pmc_sq_mimi_b32_pass$i.csv | cut -c1-200
done
