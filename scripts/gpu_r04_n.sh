# Round 4, GPU call N: split-K (2 workgroups per n-tile, fp32 partials folded by the next norm launch) for out_proj / linear_out at the
# 16-row tile, where they have one tile per CU (256 tiles): MMI_GEMM_KSPLIT=2 (the existing test hook) against the default.
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
rm -f $O/n_summary.txt
for B in 1 1 8 16; do
for ks in 1 2; do
  if [ $ks = 2 ]; then export MMI_GEMM_KSPLIT=2; else unset MMI_GEMM_KSPLIT; fi
  timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --batch $B --steps 60 --warmup 8 > $O/n_b${B}_ks$ks.log 2>&1; echo "lm B=$B mid depth, ksplit $ks: $(line $O/n_b${B}_ks$ks.log)" | tee -a $O/n_summary.txt
done
done
export MMI_GEMM_KSPLIT=2
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_n -o lm -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload lm --batch 1 --steps 40 --warmup 8 --launch-lists $O/launch_lists_n > $O/n_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_sites.py $(find /tmp/prof_n -name "*results.db" | head -1) $O/launch_lists_n --header "per-site kernel time, LMGen.step, ONE session (C3), MMI_GEMM_KSPLIT=2 (gpu_r04_n.sh)" > $O/r04_lm_b1_ks2_sites.csv 2>$O/n_sites.err
grep "^lm" $O/r04_lm_b1_ks2_sites.csv | head -26 | tee -a $O/n_summary.txt
