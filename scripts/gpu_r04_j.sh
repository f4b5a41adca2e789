# Round 4, GPU call J: one real-time session (C3).  (1) dep.ffn_in's 16-row tile at two workgroups per CU; (2) the decode attention's
# merge folded into its last workgroup + the short-ring solo path.  Parity subset first, then the B=1 A/B against the library of
# the previous commit (moshi_amd/libmoshi_mi_prev.so, built by the caller) and the B=1 site table.
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
rm -f $O/j_summary.txt
timeout 900 python -m pytest tests/test_lm_gpu.py -m gpu -x -q -k "long_ring or tiny_matches or benchmark_model_matches or full_width_layers or golden or ring_wraps or batch_rows" > $O/j_pytest.log 2>&1; echo "pytest subset: exit $? $(tail -1 $O/j_pytest.log)" | tee -a $O/j_summary.txt
for rep in 1 2; do
for lib in new prev; do
  if [ $lib = prev ]; then export MMI_LIB_PATH=$GRAFT_REPO_ROOT/moshi_amd/libmoshi_mi_prev.so; else unset MMI_LIB_PATH; fi
  timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --batch 1 --steps 60 --warmup 8 > $O/j_b1_${lib}_$rep.log 2>&1; echo "lm B=1 mid depth, $lib: $(line $O/j_b1_${lib}_$rep.log)" | tee -a $O/j_summary.txt
  [ $rep = 1 ] && timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --batch 1 --kv-depth full --steps 40 --warmup 8 > $O/j_b1full_${lib}_$rep.log 2>&1; [ $rep = 1 ] && echo "lm B=1 full context, $lib: $(line $O/j_b1full_${lib}_$rep.log)" | tee -a $O/j_summary.txt
done
done
unset MMI_LIB_PATH
for solo in 0 256 1536; do
  MMI_ATTN_SOLO=$solo timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --batch 1 --steps 60 --warmup 8 > $O/j_b1_solo$solo.log 2>&1; echo "lm B=1 mid depth, solo_rows $solo: $(line $O/j_b1_solo$solo.log)" | tee -a $O/j_summary.txt
done
timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 8 > $O/j_duplex.log 2>&1; echo "duplex B=32 (unchanged path check): $(line $O/j_duplex.log)" | tee -a $O/j_summary.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_j -o lm -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload lm --batch 1 --steps 40 --warmup 8 --launch-lists $O/launch_lists_j > $O/j_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
cp /tmp/prof_j/lm_kernel_stats.csv $O/r04_lm_b1_v2_kernel_stats.csv 2>/dev/null
python scripts/rocpd_sites.py /tmp/prof_j/lm_results.db $O/launch_lists_j --header "per-site kernel time, LMGen.step, ONE session (C3), after gpu_r04_j.sh's two changes" > $O/r04_lm_b1_v2_sites.csv 2>$O/j_sites.err
grep "^lm" $O/r04_lm_b1_v2_sites.csv | head -26 | tee -a $O/j_summary.txt
