cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_h -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --reps 3 --only hat > $GRAFT_REPO_ROOT/gpurun_out/r03aa_hat.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_export.py stats $(find /tmp/prof_h -name '*.db' | head -1) $GRAFT_REPO_ROOT/gpurun_out/r03aa_kernel_stats_hat.csv
head -14 $GRAFT_REPO_ROOT/gpurun_out/r03aa_kernel_stats_hat.csv | cut -c1-150
