# Round 4, GPU call M: k_gemm_xp at the 16-row tile bounded to 128 registers (two 8-wave workgroups per CU instead of one; 8 spilled
# registers outside the loop).  Same-box A/B against the previous commit's library, norm fusion off in both (it is a separate question).
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
rm -f $O/m_summary.txt
export MMI_NO_NORM_FUSION=1
timeout 900 python -m pytest tests/test_lm_gpu.py -m gpu -x -q -k "tiny_matches or full_width_layers or golden" > $O/m_pytest.log 2>&1; echo "pytest subset: exit $? $(tail -1 $O/m_pytest.log)" | tee -a $O/m_summary.txt
for B in 1 1 8 16; do
for lib in new prev; do
  if [ $lib = prev ]; then export MMI_LIB_PATH=$GRAFT_REPO_ROOT/moshi_amd/libmoshi_mi_prev.so; else unset MMI_LIB_PATH; fi
  timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --batch $B --steps 60 --warmup 8 > $O/m_b${B}_$lib.log 2>&1; echo "lm B=$B mid depth, $lib: $(line $O/m_b${B}_$lib.log)" | tee -a $O/m_summary.txt
done
done
unset MMI_LIB_PATH
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_m -o lm -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload lm --batch 1 --steps 40 --warmup 8 --launch-lists $O/launch_lists_m > $O/m_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_sites.py $(find /tmp/prof_m -name "*results.db" | head -1) $O/launch_lists_m --header "per-site kernel time, LMGen.step, ONE session (C3), k_gemm_xp<16> at two workgroups per CU (gpu_r04_m.sh)" > $O/r04_lm_b1_v4_sites.csv 2>$O/m_sites.err
grep "^lm,L\.\|^lm,text_lin\|^lm,dep.in_all\|TOTAL" $O/r04_lm_b1_v4_sites.csv | head -26 | tee -a $O/m_summary.txt
