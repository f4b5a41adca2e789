set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -I moshi_amd/csrc scripts/fp8_probe.hip -o /tmp/fp8_probe > /dev/null 2>&1 && timeout 60 /tmp/fp8_probe > $O/fp8_probe.log 2>&1
timeout 420 python tests/tools/fp8_gpu_diag.py bf16,fp8 wide > $O/fp8_diag.log 2>&1
cat $O/fp8_probe.log; grep -v amdgpu $O/fp8_diag.log | tail -12
