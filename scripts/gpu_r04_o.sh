# Round 4, GPU call O: the depth transformer's attention inside its out_proj (k_dep_attn_out_proj, <= 4 sessions): 42 launches fewer
# on the dependent chain.  Parity (incl. bit-identity with the two-launch form), then same-box A/B by MMI_NO_DEP_ATTN_FUSION.
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
rm -f $O/o_summary.txt
timeout 900 python -m pytest tests/test_lm_gpu.py -m gpu -x -q -k "depformer_attention_inside or tiny_matches " > $O/o_pytest.log 2>&1; echo "pytest subset: exit $? $(tail -1 $O/o_pytest.log)" | tee -a $O/o_summary.txt
for B in 1 1 2 4; do
for f in fused unfused; do
  if [ $f = unfused ]; then export MMI_NO_DEP_ATTN_FUSION=1; else unset MMI_NO_DEP_ATTN_FUSION; fi
  timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --batch $B --steps 60 --warmup 8 > $O/o_b${B}_$f.log 2>&1; echo "lm B=$B mid depth, attention $f: $(line $O/o_b${B}_$f.log)" | tee -a $O/o_summary.txt
done
done
unset MMI_NO_DEP_ATTN_FUSION
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_o -o lm -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload lm --batch 1 --steps 40 --warmup 8 --launch-lists $O/launch_lists_o > $O/o_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_o -name "*kernel_stats.csv" -exec cp {} $O/r04_lm_b1_v5_kernel_stats.csv \;
python scripts/rocpd_sites.py $(find /tmp/prof_o -name "*results.db" | head -1) $O/launch_lists_o --header "per-site kernel time, LMGen.step, ONE session (C3): attention in one launch on short rings, 16-row tiles at two workgroups per CU, depth-transformer attention inside out_proj (gpu_r04_o.sh)" > $O/r04_lm_b1_v5_sites.csv 2>$O/o_sites.err
grep "^lm" $O/r04_lm_b1_v5_sites.csv | head -26 | tee -a $O/o_summary.txt
