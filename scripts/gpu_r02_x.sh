# round 2, call x: audio-rate Mimi buffers with every row's new columns on a 128-byte line (alloc_buf) against the dense layout
# (MMI_MIMI_NO_ALIGN=1), Mimi only at 32 / 8 sessions; Mimi + batcher GPU tests; per-site times of both layouts
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_mimi_gpu.py tests/test_batcher_gpu.py -m gpu -q --timeout=600 > $O/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_subset.log
run() { # file, label, bench args..., env via VARS
  local file=$1 label=$2; shift 2
  env $VARS timeout 200 python bench.py "$@" --no-cpu-baseline > $O/ab_tmp.log 2>&1
  echo "$label $(grep '"metric"' $O/ab_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))")" >> $O/$file
}
rm -f $O/ab_align.txt $O/align_sites.txt
for rep in 1 2; do
  for b in 32 8; do
    VARS="MMI_MIMI_NO_ALIGN=1" run ab_align.txt "mimi only B=$b dense rows" --workload mimi --batch $b
    VARS="MMI_DUMMY=1" run ab_align.txt "mimi only B=$b rows on 128-byte lines" --workload mimi --batch $b
  done
done
for v in "MMI_MIMI_NO_ALIGN=1" "MMI_DUMMY=1"; do
  cd /tmp && env $v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o mimi -- python $GRAFT_REPO_ROOT/bench.py --workload mimi --no-cpu-baseline --launch-lists $O/ll_x > $O/rocprof_x.log 2>&1
  cd $GRAFT_REPO_ROOT
  echo "[$v] $(python scripts/rocpd_sites.py /tmp/prof_x/mimi_results.db $O/ll_x --header x 2>> $O/sites_err.log | grep -E 'conv0|res[0-3]|down[0-3]|convtr[0-3]|dec.final|TOTAL' | awk -F, '{printf "%s=%s ", $2, $4}')" >> $O/align_sites.txt
  rm -rf /tmp/prof_x
done
tail -3 $O/pytest_gpu_subset.log; cat $O/ab_align.txt $O/align_sites.txt
