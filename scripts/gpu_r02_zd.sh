# round 2, call zd: the depth transformer's attention launch of micro-step 0 dropped (in_proj's epilogue writes k / v into the frame cache and v
# as out_proj's operand: softmax over one position is 1) against MMI_DEP_ATTN0_LAUNCH=1; LM GPU parity subset; LM only A/B
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 400 python -m pytest tests/test_lm_gpu.py -m gpu -q --timeout=300 -x -k "greedy_schedule or 7b_layer_shapes or tiny_matches_oracle or full_width_layers or sampled_run or guidance" > $O/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_subset.log
run() { # file, label, bench args..., env via VARS
  local file=$1 label=$2; shift 2
  env $VARS timeout 100 python bench.py "$@" --no-cpu-baseline > $O/ab_tmp.log 2>&1
  echo "$label $(grep '"metric"' $O/ab_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f' % (d['ms_per_step'], d['p50_ms_per_step']))")" >> $O/$file
}
rm -f $O/ab_dep_attn0.txt
for rep in 1 2; do
  VARS="MMI_DEP_ATTN0_LAUNCH=1" run ab_dep_attn0.txt "lm only attention launch at micro-step 0" --workload lm
  VARS="MMI_DUMMY=1" run ab_dep_attn0.txt "lm only no attention launch at micro-step 0" --workload lm
done
tail -3 $O/pytest_gpu_subset.log; cat $O/ab_dep_attn0.txt
