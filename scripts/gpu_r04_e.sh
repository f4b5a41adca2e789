# Round 4, GPU call E: the attention kernel on packed dot products (same-box A/B against the chunked kernel), C5 after the
# k_gemm_q8 rework / deeper register buffers / 64 KiB LDS chunks, and site tables averaged over the post-seek steps only.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
line() { python - "$1" <<'PY'
import sys, json
try:
    d = json.loads([l for l in open(sys.argv[1]) if '"metric"' in l][-1])
    s = d.get('roofline', {}).get('sites', {})
    def g(k): return s.get(k, {}).get('us_per_op', float('nan'))
    dep = sum(v.get('us_per_step', 0) for k, v in s.items() if k.startswith('dep.') or k == 'text_sample')
    print('ms/step %.3f p50 %.3f frames/s %.0f step-frac %.3f | norm1 %.1f in_proj %.1f attn %.1f out_proj %.1f norm2 %.1f ffn_in %.1f ffn_out %.1f | dep phase (live) %.0f us' % (
        d['ms_per_step'], d.get('p50_ms_per_step', 0), d['value'], d.get('roofline', {}).get('step', {}).get('frac', 0),
        g('L.norm1'), g('L.in_proj'), g('L.attn'), g('L.out_proj'), g('L.norm2'), g('L.ffn_in'), g('L.ffn_out'), dep))
except Exception as e:
    print('no line:', e)
PY
}
rm -f $O/e_summary.txt
timeout 900 python -m pytest tests/test_lm_gpu.py -m gpu -q -x --timeout=600 -k "int8 or ring_wrap or tiny_matches or full_width_layers or fp8_kv" > $O/pytest_e.log 2>&1; echo "pytest rc=$?" | tee -a $O/e_summary.txt; tail -2 $O/pytest_e.log
for attn in split wave; do
  MMI_ATTN=$attn timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 8 > $O/e_mid_${attn}.log 2>&1
  echo "duplex depth=mid attn=$attn: $(line $O/e_mid_${attn}.log)" | tee -a $O/e_summary.txt
done
timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --kv-depth full --steps 20 --warmup 5 > $O/e_full_wave.log 2>&1; echo "lm full context: $(line $O/e_full_wave.log)" | tee -a $O/e_summary.txt
timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --batch 1 --steps 40 --warmup 8 > $O/e_lm_b1.log 2>&1; echo "lm B=1: $(line $O/e_lm_b1.log)" | tee -a $O/e_summary.txt
timeout 240 python bench.py --no-cpu-baseline --no-extras --batch 64 --quant q8 --steps 40 --warmup 8 > $O/e_b64_q8_act8.log 2>&1; echo "b64 q8 int8 activations (U=4): $(line $O/e_b64_q8_act8.log)" | tee -a $O/e_summary.txt
MMI_Q8_U=2 timeout 240 python bench.py --no-cpu-baseline --no-extras --batch 64 --quant q8 --steps 40 --warmup 8 > $O/e_b64_q8_act8_u2.log 2>&1; echo "b64 q8 int8 activations (U=2): $(line $O/e_b64_q8_act8_u2.log)" | tee -a $O/e_summary.txt
MMI_GEMM_LDS=1 timeout 240 python bench.py --no-cpu-baseline --no-extras --batch 64 --quant q8 --steps 40 --warmup 8 > $O/e_b64_q8_act8_xlds.log 2>&1; echo "b64 q8 int8 activations, k_gemm_xlds 64 KiB chunks: $(line $O/e_b64_q8_act8_xlds.log)" | tee -a $O/e_summary.txt
MMI_Q8_ACT=bf16 timeout 240 python bench.py --no-cpu-baseline --no-extras --batch 64 --quant q8 --steps 40 --warmup 8 > $O/e_b64_q8_wonly.log 2>&1; echo "b64 q8 weight-only: $(line $O/e_b64_q8_wonly.log)" | tee -a $O/e_summary.txt
# ---- rocprofv3 site tables over the post-seek steps only
cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_serial -o duplex -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --serial --launch-lists $O/launch_lists > $O/rocprof_serial.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py /tmp/prof_serial/duplex_results.db --header "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras --serial   (duplex, 32 sessions at mid-run ring depth 250 + 8 b, one stream)" > $O/r04_duplex_b32_serial_kernel_stats.csv
python scripts/rocpd_sites.py /tmp/prof_serial/duplex_results.db $O/launch_lists --header "per-site kernel time, serial schedule, 32 sessions at mid-run ring depth (250 + 8 b ... + 130): the LAST 100 steps of the trace (after the seek)" --last 100 > $O/r04_duplex_b32_serial_sites.csv
grep "^lm" $O/r04_duplex_b32_serial_sites.csv | head -24
cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_q8 -o lm -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --workload lm --batch 64 --quant q8 --launch-lists $O/launch_lists_q8 > $O/rocprof_q8.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py /tmp/prof_q8/lm_results.db --header "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras --workload lm --batch 64 --quant q8   (C5: int8 weights x int8 activations, 64 sessions)" > $O/r04_q8_b64_kernel_stats.csv
python scripts/rocpd_sites.py /tmp/prof_q8/lm_results.db $O/launch_lists_q8 --header "per-site kernel time, LMGen.step, 64 sessions at mid-run ring depth, int8 weights x int8 activations (C5): the LAST 100 steps of the trace" --last 100 > $O/r04_q8_b64_sites.csv
grep "^lm" $O/r04_q8_b64_sites.csv | head -24
