cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
# HBM traffic of the dominant kernel INSIDE the step (the product library, eager launches so that every dispatch is a packet the
# counters can be attributed to).  rocprofv3 --pmc crashed inside the profiler when attached to the Python process in rounds 1-2
# (profiles/r01_logs); tried again each round, bounded, and the standalone-launcher figure stays the committed one if it still does.
for ctr in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && MMI_NO_GRAPH=1 timeout 150 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/prof_pmc_$ctr -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload lm --steps 2 --warmup 1 > $O/rocprof_pmc_$ctr.log 2>&1; rc=$?; echo "pmc $ctr rc=$rc"
  cd $GRAFT_REPO_ROOT
  tail -3 $O/rocprof_pmc_$ctr.log | cut -c1-200
  if [ $rc != 0 ] || ! ls /tmp/prof_pmc_$ctr/*.db > /dev/null 2>&1; then break; fi
  python scripts/rocpd_pmc.py /tmp/prof_pmc_$ctr/pmc_results.db --header "rocprofv3 --pmc $ctr --kernel-trace -- MMI_NO_GRAPH=1 python bench.py --no-cpu-baseline --workload lm --steps 2 --warmup 1 (the LM step of the product library, 32 sessions)" --clusters k_gemm_xlds > $O/pmc_${ctr}_in_step.csv 2>&1
done
cat $O/pmc_FETCH_SIZE_in_step.csv | cut -c1-200 | tail -8; cat $O/pmc_WRITE_SIZE_in_step.csv | cut -c1-200 | tail -6
