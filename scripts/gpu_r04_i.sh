# Round 4, GPU call I: the duplex pipeline with the codec's two streams confined to a block of CUs (MMI_DUPLEX_CODEC_CUS=first:count).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
line() { python - "$1" <<'PY'
import sys, json
try:
    d = json.loads([l for l in open(sys.argv[1]) if '"metric"' in l][-1])
    print('ms/step %.3f p50 %.3f frames/s %.0f' % (d['ms_per_step'], d.get('p50_ms_per_step', 0), d['value']))
except Exception as e:
    print('no line:', e)
PY
}
rm -f $O/i_summary.txt
for m in none 0:64 192:64 0:128 0:32 none 128:64 0:16; do
  if [ "$m" = none ]; then unset MMI_DUPLEX_CODEC_CUS; else export MMI_DUPLEX_CODEC_CUS=$m; fi
  timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 8 > $O/i_$m.log 2>&1; echo "codec CUs $m: $(line $O/i_$m.log)" | tee -a $O/i_summary.txt
done
