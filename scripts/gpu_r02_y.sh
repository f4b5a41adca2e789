# round 2, call y: the row alignment A/B again, hoping for a box of the slow kind (where the 64-channel block that writes behind a 6-column history took 2x its twin)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rm -f $O/align_sites.txt
for v in "MMI_MIMI_NO_ALIGN=1" "MMI_DUMMY=1" "MMI_MIMI_NO_RES_FUSION=1"; do
  cd /tmp && env $v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o mimi -- python $GRAFT_REPO_ROOT/bench.py --workload mimi --no-cpu-baseline --launch-lists $O/ll_x > $O/rocprof_x.log 2>&1
  cd $GRAFT_REPO_ROOT
  echo "[$v] $(grep '"metric"' $O/rocprof_x.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f' % d['ms_per_step'])") $(python scripts/rocpd_sites.py /tmp/prof_x/mimi_results.db $O/ll_x --header x 2>> $O/sites_err.log | grep -E 'conv0|res[0-3]|down[0-3]|convtr[0-3]|dec.final|TOTAL' | awk -F, '{printf "%s=%s ", $2, $4}')" >> $O/align_sites.txt
  rm -rf /tmp/prof_x
done
cat $O/align_sites.txt
