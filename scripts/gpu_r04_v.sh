# Round 4, GPU call V: the same crossover (gpu_r04_u.sh) at 2 / 4 / 8 / 16 sessions and deep rings.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
rm -f $O/v_summary.txt
for B in 2 4 8 16; do
for depth in 1800 3000; do
for solo in 0 100000; do
  MMI_ATTN_SOLO=$solo timeout 100 python bench.py --no-cpu-baseline --no-extras --workload lm --batch $B --kv-seek $depth --steps 30 --warmup 6 > $O/v_b${B}_${depth}_$solo.log 2>&1
  echo "lm B=$B depth $depth solo_rows $solo: $(grep '"metric"' $O/v_b${B}_${depth}_$solo.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))")" | tee -a $O/v_summary.txt
done
done
done
