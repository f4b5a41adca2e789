# Counter passes over the LM step on the product library, eager launches (every dispatch is a packet the counters can be
# attributed to).  gpurun -- 'bash scripts/gpu_pmc_step.sh'.  VERDICT r4 items 3 and 7:
#  * 32 sessions, bf16, mid-run ring depth: FETCH_SIZE / WRITE_SIZE / TCC for the dominant GEMM (k_gemm_xlds) AND the decode attention
#    (k_lm_attn_wave), so that roofline.traffic is a figure of THIS round and the attention's clamped re-reads are counted;
#  * the same FETCH_SIZE pass with every ring 3000 deep (--kv-depth full);
#  * 64 sessions, int8 x int8 (C5): SQ (busy / wait / MFMA-busy / VALU / LDS) and TCC passes over k_gemm_xp<.., WQ = 3>, and the SQ
#    pass over the bf16 kernels at the same batch beside it.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
run() {   # name, counters, bench args...
  local name=$1 ctrs=$2; shift 2
  ( cd /tmp && MMI_NO_GRAPH=1 timeout 200 rocprofv3 --pmc $ctrs --kernel-trace -d /tmp/pmc_$name -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --workload lm --steps 2 --warmup 1 "$@" > $O/pmc_$name.log 2>&1 ); echo "$name rc=$?"
}
post() {  # name, header, clusters-substring, extra flags
  PMC_ROWS=60 python scripts/rocpd_pmc.py /tmp/pmc_$1/pmc_results.db --header "$2" --clusters "$3" $4 > $O/r05_pmc_$1.csv 2>&1
}
HDR="MMI_NO_GRAPH=1 python bench.py --no-cpu-baseline --no-extras --workload lm --steps 2 --warmup 1"
for ctr in FETCH_SIZE WRITE_SIZE; do
  run b32_$ctr $ctr --batch 32
  post b32_$ctr "rocprofv3 --pmc $ctr --kernel-trace -- $HDR --batch 32 (bf16, mid-run ring depth 250 + 8 b)" "k_" ""
done
run b32_TCC "TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum" --batch 32
post b32_TCC "rocprofv3 --pmc TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum -- $HDR --batch 32 (raw counter values in the avg_KiB column)" "k_" "--by-duration"
run b32_full_FETCH_SIZE FETCH_SIZE --batch 32 --kv-depth full
post b32_full_FETCH_SIZE "rocprofv3 --pmc FETCH_SIZE -- $HDR --batch 32 --kv-depth full (every ring 3000 deep)" "k_lm_attn" ""
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY"
run q8_b64_SQ "$SQ" --batch 64 --quant q8
post q8_b64_SQ "rocprofv3 --pmc $SQ -- $HDR --batch 64 --quant q8 (int8 x int8; raw counter values in the avg_KiB column)" "k_gemm" "--by-duration"
run q8_b64_TCC "TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum" --batch 64 --quant q8
post q8_b64_TCC "rocprofv3 --pmc TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum -- $HDR --batch 64 --quant q8 (raw counter values in the avg_KiB column)" "k_gemm" "--by-duration"
run bf16_b64_SQ "$SQ" --batch 64
post bf16_b64_SQ "rocprofv3 --pmc $SQ -- $HDR --batch 64 (bf16; raw counter values in the avg_KiB column)" "k_gemm" "--by-duration"
for f in $O/r05_pmc_*.csv; do echo "== $f"; grep -A40 "clusters" $f | cut -c1-170 | head -24; done
