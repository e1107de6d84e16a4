cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_d -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --reps 3 --only drct > $GRAFT_REPO_ROOT/gpurun_out/r03aa_drct.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_export.py stats $(find /tmp/prof_d -name '*.db' | head -1) $GRAFT_REPO_ROOT/gpurun_out/r03aa_kernel_stats_drct.csv
head -14 $GRAFT_REPO_ROOT/gpurun_out/r03aa_kernel_stats_drct.csv | cut -c1-150
