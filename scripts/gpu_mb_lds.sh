set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -Imoshi_amd/csrc scripts/gemm_microbench.hip moshi_amd/csrc/api_common.hip -o /tmp/gemm_microbench > $O/mb_build.log 2>&1 || cat $O/mb_build.log
MB_LDS=1 timeout 120 /tmp/gemm_microbench 32 1 quick > $O/mb_lds_b32.txt 2>&1
MB_LDS=1 timeout 120 /tmp/gemm_microbench 64 1 quick > $O/mb_lds_b64.txt 2>&1
cat $O/mb_lds_b32.txt $O/mb_lds_b64.txt
