set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -Imoshi_amd/csrc scripts/gemm_microbench.hip moshi_amd/csrc/api_common.hip -o /tmp/gemm_microbench > $O/mb_build.log 2>&1
MB_T=16 timeout 300 /tmp/gemm_microbench 32 1 > $O/gemm_microbench_b32_t16.txt 2>&1
timeout 300 /tmp/gemm_microbench 32 1 > $O/gemm_microbench_b32_t32.txt 2>&1
echo T16; grep -E "^== dep|16x2" $O/gemm_microbench_b32_t16.txt | tail -24
echo T32; grep -E "^== dep|32x1 ntw1 w8 u4|32x1 ntw1 w4 u4" $O/gemm_microbench_b32_t32.txt | tail -12
