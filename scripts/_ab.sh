cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
line() { python -c "
import sys,json
d=json.loads([l for l in open('$1') if '\"metric\"' in l][-1]); s=d['roofline']['sites']
print('ms/step %.3f p50 %.3f frames/s %.0f | in_proj %.2f ffn_in %.2f out_proj %.2f ffn_out %.2f' % (d['ms_per_step'], d['p50_ms_per_step'], d['value'], s['L.in_proj']['us_per_op'], s['L.ffn_in']['us_per_op'], s['L.out_proj']['us_per_op'], s['L.ffn_out']['us_per_op']))"; }
rm -f $O/q8w_summary.txt
for w in 8 4 8 4; do
  MMI_Q8_WAVES=$w timeout 60 python bench.py --no-cpu-baseline --batch 64 --quant q8 --steps 30 --warmup 8 > $O/q8w_b.log 2>&1; echo "b64 q8 waves=$w: $(line $O/q8w_b.log)" | tee -a $O/q8w_summary.txt
done
MMI_Q8_WAVES=4 timeout 90 python -m pytest tests/test_lm_gpu.py -m gpu -q -x -k "c5_shape_int8" > $O/q8w_test.log 2>&1; echo "c5 int8 test with 4 waves rc=$?" | tee -a $O/q8w_summary.txt; tail -2 $O/q8w_test.log
