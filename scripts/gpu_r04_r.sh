# Round 4, GPU call R: k_gemm_xlds with THREE weight register buffers (two segments in flight behind the one on the matrix core) at
# 32 sessions.  Parity of the GEMMs that take it, then same-box A/B against the previous commit's library (moshi_amd/libmoshi_mi_prev.so),
# staggered tails (default; 13 spilled registers in the new form) and plain tails (MMI_GEMM_LDS=1; none).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
line() { python - "$1" <<'PY'
import sys, json
try:
    d = json.loads([l for l in open(sys.argv[1]) if '"metric"' in l][-1])
    r = d.get('roofline', {})
    print('ms/step %.3f p50 %.3f dominant %.2f us' % (d['ms_per_step'], d.get('p50_ms_per_step', 0), r.get('kernel_us', 0) or 0))
except Exception as e:
    print('no line:', e)
PY
}
rm -f $O/r_summary.txt
timeout 900 python -m pytest tests/test_lm_gpu.py -m gpu -x -q -k "lds_resident or benchmark_model_matches or full_depth_32 or batch_rows or two_batch_tiles or golden" > $O/r_pytest.log 2>&1; echo "pytest subset: exit $? $(tail -1 $O/r_pytest.log)" | tee -a $O/r_summary.txt
for rep in 1 2; do
for lib in new prev; do
for lds in default 1; do
  if [ $lib = prev ]; then export MMI_LIB_PATH=$GRAFT_REPO_ROOT/moshi_amd/libmoshi_mi_prev.so; else unset MMI_LIB_PATH; fi
  if [ $lds = 1 ]; then export MMI_GEMM_LDS=1; else unset MMI_GEMM_LDS; fi
  timeout 200 python bench.py --no-cpu-baseline --no-extras --serial --steps 40 --warmup 8 > $O/r_${lib}_${lds}_$rep.log 2>&1; echo "duplex B=32 serial, lib $lib, MMI_GEMM_LDS $lds: $(line $O/r_${lib}_${lds}_$rep.log)" | tee -a $O/r_summary.txt
done
done
done
unset MMI_GEMM_LDS
for lib in new prev; do
  if [ $lib = prev ]; then export MMI_LIB_PATH=$GRAFT_REPO_ROOT/moshi_amd/libmoshi_mi_prev.so; else unset MMI_LIB_PATH; fi
  timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 8 > $O/r_pipe_$lib.log 2>&1; echo "duplex B=32 pipelined, lib $lib: $(line $O/r_pipe_$lib.log)" | tee -a $O/r_summary.txt
done
unset MMI_LIB_PATH
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_r -o d -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --serial --steps 40 --warmup 8 --launch-lists $O/launch_lists_r > $O/r_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_r -name "*kernel_stats.csv" -exec cp {} $O/r04_duplex_b32_serial_v3_kernel_stats.csv \;
python scripts/rocpd_sites.py $(find /tmp/prof_r -name "*results.db" | head -1) $O/launch_lists_r --last 100 --header "per-site kernel time, python bench.py --serial (32 sessions, mid-run depth), k_gemm_xlds with three weight buffers (gpu_r04_r.sh)" > $O/r04_duplex_b32_serial_v3_sites.csv 2>$O/r_sites.err
grep "^lm,L\.\|^lm,text_l\|TOTAL" $O/r04_duplex_b32_serial_v3_sites.csv | head -12 | tee -a $O/r_summary.txt
