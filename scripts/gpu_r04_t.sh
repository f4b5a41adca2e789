# Round 4, GPU call T: LMGen.step alone at 2 / 4 / 8 / 16 sessions (mid-run ring depth), for the sessions-vs-latency table of DESIGN 6.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
rm -f $O/t_summary.txt
for B in 2 4 8 16; do
  timeout 100 python bench.py --no-cpu-baseline --no-extras --workload lm --batch $B --steps 40 --warmup 8 > $O/t_b$B.log 2>&1
  echo "lm B=$B: $(grep '"metric"' $O/t_b$B.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f frames/s %.0f' % (d['ms_per_step'], d['p50_ms_per_step'], d['value']))")" | tee -a $O/t_summary.txt
done
