cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
line() { python -c "
import sys,json
d=json.loads([l for l in open('$1') if '\"metric\"' in l][-1]); s=d['roofline']['sites']
print('ms/step %.3f p50 %.3f frames/s %.0f | norm1 %.2f in_proj %.2f norm2 %.2f ffn_in %.2f out_proj %.2f ffn_out %.2f attn %.2f' % (d['ms_per_step'], d['p50_ms_per_step'], d['value'], s['L.norm1']['us_per_op'], s['L.in_proj']['us_per_op'], s['L.norm2']['us_per_op'], s['L.ffn_in']['us_per_op'], s['L.out_proj']['us_per_op'], s['L.ffn_out']['us_per_op'], s['L.attn']['us_per_op']))"; }
rm -f $O/z_summary.txt
for rep in 1 2; do
for v in off 8 16 24 36; do
  if [ $v = off ]; then export MMI_NO_NORM_PREFETCH=1; unset MMI_NORM_PREFETCH_MB; else unset MMI_NO_NORM_PREFETCH; export MMI_NORM_PREFETCH_MB=$v; fi
  timeout 200 python bench.py --no-cpu-baseline --serial > $O/z_b.log 2>&1; echo "serial prefetch=$v: $(line $O/z_b.log)" | tee -a $O/z_summary.txt
  timeout 200 python bench.py --no-cpu-baseline > $O/z_b.log 2>&1; echo "pipelined prefetch=$v: $(line $O/z_b.log)" | tee -a $O/z_summary.txt
done; done
