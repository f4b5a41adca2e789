# Round 4, GPU call L: RMSNorm fused into the temporal layers' 16-row GEMMs (k_gemm_norm16: norm1 -> in_proj, norm2 -> linear_in;
# 64 launches fewer per step at <= 16 sessions).  Parity (incl. bit-identity with the two-launch form), then same-box A/B.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
line() { python - "$1" <<'PY'
import sys, json
try:
    d = json.loads([l for l in open(sys.argv[1]) if '"metric"' in l][-1])
    print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d.get('p50_ms_per_step', 0)))
except Exception as e:
    print('no line:', e)
PY
}
rm -f $O/l_summary.txt
timeout 900 python -m pytest tests/test_lm_gpu.py -m gpu -x -q -k "norm_fused or tiny_matches" > $O/l_pytest.log 2>&1; echo "pytest subset: exit $? $(tail -1 $O/l_pytest.log)" | tee -a $O/l_summary.txt
for rep in 1 2; do
for nf in fused unfused; do
  if [ $nf = unfused ]; then export MMI_NO_NORM_FUSION=1; else unset MMI_NO_NORM_FUSION; fi
  timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --batch 1 --steps 60 --warmup 8 > $O/l_b1_${nf}_$rep.log 2>&1; echo "lm B=1 mid depth, $nf: $(line $O/l_b1_${nf}_$rep.log)" | tee -a $O/l_summary.txt
done
done
for B in 8 16; do
for nf in fused unfused; do
  if [ $nf = unfused ]; then export MMI_NO_NORM_FUSION=1; else unset MMI_NO_NORM_FUSION; fi
  timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --batch $B --steps 40 --warmup 8 > $O/l_b${B}_$nf.log 2>&1; echo "lm B=$B mid depth, $nf: $(line $O/l_b${B}_$nf.log)" | tee -a $O/l_summary.txt
done
done
unset MMI_NO_NORM_FUSION
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_l -o lm -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload lm --batch 1 --steps 40 --warmup 8 --launch-lists $O/launch_lists_l > $O/l_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_l -name "*kernel_stats.csv" -exec cp {} $O/r04_lm_b1_v3_kernel_stats.csv \;
python scripts/rocpd_sites.py $(find /tmp/prof_l -name "*results.db" | head -1) $O/launch_lists_l --header "per-site kernel time, LMGen.step, ONE session (C3), norms fused into the 16-row GEMMs (gpu_r04_l.sh)" > $O/r04_lm_b1_v3_sites.csv 2>$O/l_sites.err
grep "^lm" $O/r04_lm_b1_v3_sites.csv | head -26 | tee -a $O/l_summary.txt
