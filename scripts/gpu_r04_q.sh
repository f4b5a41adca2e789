# Round 4, GPU call Q: at 32 sessions, k_gemm_xp at the 32-row tile bounded to 128 registers (two workgroups per CU, 14 spilled
# registers in the epilogue; moshi_amd/libmoshi_mi_exp.so, built by the caller) - on its own and against k_gemm_xlds for the temporal
# in_proj / linear_in / text head (MMI_GEMM_LDS=0 puts them on k_gemm_xp).
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
rm -f $O/q_summary.txt
for rep in 1 2; do
for lib in cur exp; do
for lds in default 0; do
  if [ $lib = exp ]; then export MMI_LIB_PATH=$GRAFT_REPO_ROOT/moshi_amd/libmoshi_mi_exp.so; else unset MMI_LIB_PATH; fi
  if [ $lds = 0 ]; then export MMI_GEMM_LDS=0; else unset MMI_GEMM_LDS; fi
  timeout 200 python bench.py --no-cpu-baseline --no-extras --serial --steps 40 --warmup 8 > $O/q_${lib}_${lds}_$rep.log 2>&1; echo "duplex B=32 serial, lib $lib, MMI_GEMM_LDS $lds: $(line $O/q_${lib}_${lds}_$rep.log)" | tee -a $O/q_summary.txt
done
done
done
