# round 3, call p: several independent session groups per GPU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
for a in "--groups 1" "--groups 2" "--groups 2 --serial" "--groups 3" "--groups 1 --batch 64" "--groups 2 --quant q8" "--groups 1 --batch 64 --quant q8"; do
  timeout 300 python scripts/two_groups.py $a 2>&1 | grep -v amdgpu.ids | tee -a $O/p_groups.txt
done
