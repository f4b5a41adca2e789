# Round 4, GPU call K: call J's in-kernel merge was 8 us per layer slower than the k_lm_attn_combine launch on deep rings, so the
# engine now keeps two step programs (lm_engine.hip attn_variant).  Parity of all the paths, then same-box A/B against the
# previous commit's library (moshi_amd/libmoshi_mi_prev.so, built by the caller) at one session and at 8.
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
rm -f $O/k_summary.txt
timeout 900 python -m pytest tests/test_lm_gpu.py -m gpu -x -q -k "long_ring or tiny_matches or benchmark_model_matches or golden or ring_wraps or batch_rows or resume" > $O/k_pytest.log 2>&1; echo "pytest subset: exit $? $(tail -1 $O/k_pytest.log)" | tee -a $O/k_summary.txt
for lib in new prev new prev; do
  if [ $lib = prev ]; then export MMI_LIB_PATH=$GRAFT_REPO_ROOT/moshi_amd/libmoshi_mi_prev.so; else unset MMI_LIB_PATH; fi
  timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --batch 1 --steps 60 --warmup 8 > $O/k_b1_$lib.log 2>&1; echo "lm B=1 mid depth, $lib: $(line $O/k_b1_$lib.log)" | tee -a $O/k_summary.txt
  timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --batch 1 --kv-depth full --steps 40 --warmup 8 > $O/k_b1full_$lib.log 2>&1; echo "lm B=1 full context, $lib: $(line $O/k_b1full_$lib.log)" | tee -a $O/k_summary.txt
done
for lib in new prev; do
  if [ $lib = prev ]; then export MMI_LIB_PATH=$GRAFT_REPO_ROOT/moshi_amd/libmoshi_mi_prev.so; else unset MMI_LIB_PATH; fi
  timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --batch 8 --steps 40 --warmup 8 > $O/k_b8_$lib.log 2>&1; echo "lm B=8 mid depth, $lib: $(line $O/k_b8_$lib.log)" | tee -a $O/k_summary.txt
  timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --batch 8 --kv-depth full --steps 40 --warmup 8 > $O/k_b8full_$lib.log 2>&1; echo "lm B=8 full context, $lib: $(line $O/k_b8full_$lib.log)" | tee -a $O/k_summary.txt
done
unset MMI_LIB_PATH
MMI_ATTN_SOLO=1536 timeout 200 python bench.py --no-cpu-baseline --no-extras --workload lm --batch 1 --kv-depth full --steps 40 --warmup 8 > $O/k_b1full_solo.log 2>&1; echo "lm B=1, ring 1536 deep... (full, solo_rows 1536 - only shows the program switch is depth driven): $(line $O/k_b1full_solo.log)" | tee -a $O/k_summary.txt
timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 8 > $O/k_duplex.log 2>&1; echo "duplex B=32 (unchanged path check): $(line $O/k_duplex.log)" | tee -a $O/k_summary.txt
