# round 2, sixth GPU call: HBM traffic (PMC) of the new dominant kernel k_gemm_xlds on the standalone launcher (separate passes,
# --kernel-trace only), and the other configurations' lines for the record (C3, C5, served, full context)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out build
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -Imoshi_amd/csrc scripts/gemm_microbench.hip moshi_amd/csrc/api_common.hip -o build/gemm_microbench > $O/mb_build.log 2>&1 || cat $O/mb_build.log
MB_LDS=1 timeout 200 build/gemm_microbench 32 1 quick > $O/mb_lds_b32_quick.txt 2>&1
cd /tmp && MB_LDS=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o pmc -- $GRAFT_REPO_ROOT/build/gemm_microbench 32 1 quick > $O/pmc_fetch.log 2>&1
cd /tmp && MB_LDS=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o pmc -- $GRAFT_REPO_ROOT/build/gemm_microbench 32 1 quick > $O/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_pmc.py $O/pmc_fetch/pmc_results.db --header "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- MB_LDS=1 build/gemm_microbench 32 1 quick" > $O/r02_pmc_fetch_ffn_in_xlds.csv 2>> $O/pmc_fetch.log
python scripts/rocpd_pmc.py $O/pmc_write/pmc_results.db --header "rocprofv3 --pmc WRITE_SIZE --kernel-trace -- MB_LDS=1 build/gemm_microbench 32 1 quick" > $O/r02_pmc_write_ffn_in_xlds.csv 2>> $O/pmc_write.log
rm -rf $O/pmc_fetch $O/pmc_write
timeout 200 python bench.py --workload lm --batch 1 --no-cpu-baseline > $O/bench_lm_b1.log 2>&1
timeout 300 python bench.py --batch 64 --quant q8 --no-cpu-baseline > $O/bench_duplex_b64_q8.log 2>&1
timeout 300 python bench.py --workload served --no-cpu-baseline > $O/bench_served_b32.log 2>&1
cat $O/mb_lds_b32_quick.txt | tail -12; cat $O/r02_pmc_fetch_ffn_in_xlds.csv $O/r02_pmc_write_ffn_in_xlds.csv | cut -c1-220
for f in bench_lm_b1 bench_duplex_b64_q8 bench_served_b32; do grep '"metric"' $O/$f.log | cut -c1-330; done
