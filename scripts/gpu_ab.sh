# A/B on ONE box: the commit this session started from (ab_old/, 5548462) against the current tree, default benchmark command
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
rocm-smi --showclocks --showpower --showperflevel > $O/rocm_smi_before.log 2>&1
( cd ab_old && timeout 400 python bench.py --no-cpu-baseline ) > $O/ab_old_bench.log 2>&1
timeout 400 python bench.py --no-cpu-baseline > $O/ab_new_bench.log 2>&1
( cd ab_old && timeout 400 python bench.py --no-cpu-baseline ) > $O/ab_old_bench2.log 2>&1
timeout 400 python bench.py --no-cpu-baseline > $O/ab_new_bench2.log 2>&1
rocm-smi --showclocks --showpower > $O/rocm_smi_after.log 2>&1
for f in ab_old_bench ab_new_bench ab_old_bench2 ab_new_bench2; do echo $f; grep '"metric"' $O/$f.log | cut -c1-260; done
grep -iE "sclk|mclk|power|perf" $O/rocm_smi_before.log | head -12
