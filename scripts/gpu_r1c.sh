set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -I moshi_amd/csrc scripts/fp8_probe.hip -o /tmp/fp8_probe > $O/fp8_probe.log 2>&1 && timeout 60 /tmp/fp8_probe >> $O/fp8_probe.log 2>&1
timeout 300 python tests/tools/fp8_gpu_diag.py > $O/fp8_diag.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -v warning $O/fp8_probe.log | tail -8; cat $O/fp8_diag.log | tail -12; tail -n 12 $O/pytest_gpu.log
