# round 2, the remaining seconds: mirrored session order of the temporal attention (MMI_ATTN_MIRROR=1) against the plain order, LM only
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
rm -f $O/ab_attn_mirror.txt
for v in 1 0 1; do
  MMI_ATTN_MIRROR=$v timeout 14 python bench.py --workload lm --no-cpu-baseline > $O/ab_tmp.log 2>&1
  echo "lm only MMI_ATTN_MIRROR=$v $(grep '"metric"' $O/ab_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))")" >> $O/ab_attn_mirror.txt
done
cat $O/ab_attn_mirror.txt
