# The end-of-round GPU check when little of the round's GPU budget is left (gpurun -- 'bash scripts/gpu_check_short.sh'): every -m gpu
# test (two xdist workers on the one GPU), smoke, the default benchmark line as the driver runs it, the one-session line.  The other
# named configurations (64 sessions; Mimi alone) are in scripts/gpu_check.sh; nothing on their paths changed after that check ran.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -n 2 --timeout=900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.log 2>&1; grep '"metric"' $O/bench_default.log | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline --workload lm --batch 1 > $O/bench_lm_b1.log 2>&1
rm -f $O/bench_lines.txt
for f in bench_default bench_lm_b1; do
  echo "$f: $(grep '"metric"' $O/$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f frames/s %.0f' % (d['ms_per_step'], d['p50_ms_per_step'], d['value']))")" | tee -a $O/bench_lines.txt
done
grep '"metric"' $O/bench_default.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
for k in ('kv_depth_start', 'full_context'):
    print(k, json.dumps(d.get(k))[:300])
c3 = dict(d.get('c3') or {}); c3.pop('sites', None); print('c3', json.dumps(c3)[:400])
print('roofline', json.dumps({k: v for k, v in d['roofline'].items() if k != 'sites'})[:600])
print('cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:300])" | tee -a $O/bench_lines.txt
