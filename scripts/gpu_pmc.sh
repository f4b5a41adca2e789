set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -o pmc -- $GRAFT_REPO_ROOT/build/gemm_microbench 32 1 quick > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log 2>&1
cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -o pmc -- $GRAFT_REPO_ROOT/build/gemm_microbench 32 1 quick > $GRAFT_REPO_ROOT/gpurun_out/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/pmc_fetch gpurun_out/pmc_write; tail -5 gpurun_out/pmc_fetch.log
