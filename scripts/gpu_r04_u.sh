# Round 4, GPU call U: where the one-launch decode attention (workgroup 0 alone) stops paying against ring split + merge launch:
# one session moved to ring depths 300 ... 2400, MMI_ATTN_SOLO=0 (always the merge launch) against 100000 (always alone).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
rm -f $O/u_summary.txt
for depth in 300 600 900 1200 1600 2400; do
for solo in 0 100000; do
  MMI_ATTN_SOLO=$solo timeout 100 python bench.py --no-cpu-baseline --no-extras --workload lm --batch 1 --kv-seek $depth --steps 40 --warmup 8 > $O/u_${depth}_$solo.log 2>&1
  echo "lm B=1 depth $depth solo_rows $solo: $(grep '"metric"' $O/u_${depth}_$solo.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))")" | tee -a $O/u_summary.txt
done
done
for B in 4 8; do
for depth in 600 1200; do
for solo in 0 100000; do
  MMI_ATTN_SOLO=$solo timeout 100 python bench.py --no-cpu-baseline --no-extras --workload lm --batch $B --kv-seek $depth --steps 40 --warmup 8 > $O/u_b${B}_${depth}_$solo.log 2>&1
  echo "lm B=$B depth $depth solo_rows $solo: $(grep '"metric"' $O/u_b${B}_${depth}_$solo.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))")" | tee -a $O/u_summary.txt
done
done
done
