set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out build
O=$GRAFT_REPO_ROOT/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -Imoshi_amd/csrc scripts/gemm_microbench.hip moshi_amd/csrc/api_common.hip -o /tmp/gemm_microbench > $O/mb_build.log 2>&1
timeout 300 /tmp/gemm_microbench 64 1 > $O/gemm_microbench_b64.txt 2>&1
timeout 300 /tmp/gemm_microbench 64 2 > $O/gemm_microbench_b64_ksplit2.txt 2>&1
grep -E "^==|32x2" $O/gemm_microbench_b64.txt | head -60
