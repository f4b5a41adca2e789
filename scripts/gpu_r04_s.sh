# Round 4, GPU call S: the one-session step as committed - rocprofv3 kernel stats + per-site table from ONE run (profiles/r04_lm_b1_v5_*).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o lm -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload lm --batch 1 --steps 40 --warmup 8 --launch-lists $O/launch_lists_s > $O/s_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_s -name "*results.db" | head -1)
python scripts/rocpd_stats.py $DB --header "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --workload lm --batch 1   (C3: Moshi-7B LMGen.step, one session, end of round 4)" > $O/r04_lm_b1_v5_kernel_stats.csv
python scripts/rocpd_sites.py $DB $O/launch_lists_s --header "per-site kernel time, LMGen.step, ONE session (C3), end of round 4: attention in one launch on short rings, 16-row tiles at two workgroups per CU, depth-transformer attention inside out_proj (gpu_r04_s.sh)" > $O/r04_lm_b1_v5_sites.csv 2>$O/s_sites.err
grep '"metric"' $O/s_rocprof.log | cut -c1-200; grep "TOTAL" $O/r04_lm_b1_v5_sites.csv; head -8 $O/r04_lm_b1_v5_kernel_stats.csv | cut -c1-160
