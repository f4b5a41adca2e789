# Round 4, GPU call P: k_gemm_xp ONESHOT (a wave's whole K-slice in one request) for the GEMMs of <= 256 workgroups with 5-23 k-steps
# per wave.  Parity, then same-box A/B by MMI_GEMM_ONESHOT=0 at 1 / 8 / 32 (pipelined default) / 64 sessions, and the B = 32 site table.
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
rm -f $O/p_summary.txt
timeout 900 python -m pytest tests/test_lm_gpu.py -m gpu -x -q -k "one_request or tiny_matches or golden or full_width_layers or benchmark_model_matches" > $O/p_pytest.log 2>&1; echo "pytest subset: exit $? $(tail -1 $O/p_pytest.log)" | tee -a $O/p_summary.txt
for rep in 1 2; do
for m in on off; do
  if [ $m = off ]; then export MMI_GEMM_ONESHOT=0; else unset MMI_GEMM_ONESHOT; fi
  timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --batch 1 --steps 60 --warmup 8 > $O/p_b1_${m}_$rep.log 2>&1; echo "lm B=1, oneshot $m: $(line $O/p_b1_${m}_$rep.log)" | tee -a $O/p_summary.txt
  timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 8 > $O/p_b32_${m}_$rep.log 2>&1; echo "duplex B=32, oneshot $m: $(line $O/p_b32_${m}_$rep.log)" | tee -a $O/p_summary.txt
done
done
for m in on off; do
  if [ $m = off ]; then export MMI_GEMM_ONESHOT=0; else unset MMI_GEMM_ONESHOT; fi
  timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --batch 8 --steps 40 --warmup 8 > $O/p_b8_$m.log 2>&1; echo "lm B=8, oneshot $m: $(line $O/p_b8_$m.log)" | tee -a $O/p_summary.txt
  timeout 200 python bench.py --no-cpu-baseline --no-extras --serial --steps 40 --warmup 8 > $O/p_b32s_$m.log 2>&1; echo "duplex B=32 serial, oneshot $m: $(line $O/p_b32s_$m.log)" | tee -a $O/p_summary.txt
done
unset MMI_GEMM_ONESHOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o d -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --serial --steps 120 --warmup 8 --launch-lists $O/launch_lists_p > $O/p_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_p -name "*kernel_stats.csv" -exec cp {} $O/r04_duplex_b32_serial_v2_kernel_stats.csv \;
python scripts/rocpd_sites.py $(find /tmp/prof_p -name "*results.db" | head -1) $O/launch_lists_p --last 100 --header "per-site kernel time, python bench.py --serial (32 sessions, mid-run depth), after k_gemm_xp ONESHOT (gpu_r04_p.sh)" > $O/r04_duplex_b32_serial_v2_sites.csv 2>$O/p_sites.err
grep "^lm" $O/r04_duplex_b32_serial_v2_sites.csv | head -26 | tee -a $O/p_summary.txt
