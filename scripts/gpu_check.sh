# The round's GPU check on one MI355X box (gpurun -- 'bash scripts/gpu_check.sh'): smoke, every -m gpu test, the default benchmark
# line as the driver runs it (with its extras and its CPU-baseline leg), and the other named configurations (BASELINE.json
# configs 1, 2, 4) as logged lines.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
python scripts/isa_scan_packed_swizzle.py moshi_amd/libmoshi_mi.so | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
# exactly as the driver runs it (-x: the first failure stops the run), plus the slowest tests for the suite's time budget (600 s)
timeout 1500 python -m pytest tests -x -q -m gpu --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -18 $O/pytest_gpu.log | cut -c1-200
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.log 2>&1; grep '"metric"' $O/bench_default.log | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline --no-extras --serial > $O/bench_default_serial.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --workload mimi --batch 8 > $O/bench_mimi_b8.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --workload lm --batch 1 > $O/bench_lm_b1.log 2>&1
for q in none q8 fp8; do timeout 300 python bench.py --no-cpu-baseline --no-extras --batch 64 --quant $q > $O/bench_duplex_b64_$q.log 2>&1; done
MMI_Q8_ACT=bf16 timeout 300 python bench.py --no-cpu-baseline --no-extras --batch 64 --quant q8 > $O/bench_duplex_b64_q8_weight_only.log 2>&1
rm -f $O/bench_lines.txt
for f in bench_default bench_default_serial bench_mimi_b8 bench_lm_b1 bench_duplex_b64_none bench_duplex_b64_q8 bench_duplex_b64_q8_weight_only bench_duplex_b64_fp8; do
  echo "$f: $(grep '"metric"' $O/$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f frames/s %.0f' % (d['ms_per_step'], d['p50_ms_per_step'], d['value']))")" | tee -a $O/bench_lines.txt
done
grep '"metric"' $O/bench_default.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
for k in ('kv_depth_start', 'full_context'):
    print(k, json.dumps(d.get(k))[:300])
c3 = dict(d.get('c3') or {}); c3.pop('sites', None); print('c3', json.dumps(c3)[:400])
print('cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:300])" | tee -a $O/bench_lines.txt
