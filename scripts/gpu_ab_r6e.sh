# stream priorities of the duplex pipeline, three interleaved pairs on one box (MMI_DUPLEX_PRIO: unset = codec streams high, LM low; lm = the reverse)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
rm -f $O/q_*.log $O/q_lines.txt
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f value %.0f' % (d['ms_per_step'], d['p50_ms_per_step'], d['value']))"; }
run() { name=$1; shift; ( env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extras $BARGS ) > $O/q_$name.log 2>&1; echo "$name: $(line $O/q_$name.log)" | tee -a $O/q_lines.txt; }
BARGS=""
for i in 1 2 3; do
  run codec_high_$i X=1
  run lm_high_$i MMI_DUPLEX_PRIO=lm
done
BARGS="--quant q8 --kv fp8 --batch 64"
run c5_codec_high X=1
run c5_lm_high MMI_DUPLEX_PRIO=lm
BARGS="--batch 8"
run b8_codec_high X=1
run b8_lm_high MMI_DUPLEX_PRIO=lm
