# SQ / TCC counters of the FFN linear_in GEMM with one (32 sessions) and two (64 sessions) batch tiles per weight fragment:
# where do the extra 9-12 us of the second tile go?  (separate --pmc passes, --kernel-trace only)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -Imoshi_amd/csrc scripts/gemm_microbench.hip moshi_amd/csrc/api_common.hip -o /tmp/gemm_microbench > $O/mb_build.log 2>&1
for B in 32 64; do
  cd /tmp && timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /tmp/pmc_sq_$B -o pmc -- /tmp/gemm_microbench $B 1 quick > $O/pmc_sq_$B.log 2>&1
  cd /tmp && timeout 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d /tmp/pmc_tcc_$B -o pmc -- /tmp/gemm_microbench $B 1 quick > $O/pmc_tcc_$B.log 2>&1
  cd $GRAFT_REPO_ROOT
  for d in /tmp/pmc_sq_$B /tmp/pmc_tcc_$B; do
    db=$(find $d -name "*.db" | head -1)
    python scripts/rocpd_pmc.py $db | grep -E "k_gemm_xp|kernel," | cut -c1-200 >> $O/pmc_sq_summary_b$B.csv
  done
done
cat $O/pmc_sq_summary_b32.csv; cat $O/pmc_sq_summary_b64.csv
