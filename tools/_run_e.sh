cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_conv_fp16_gpu.py tests/test_conv_gpu.py tests/test_rrdbnet_gpu.py tests/test_baseline_configs_gpu.py -x -q -m gpu > gpurun_out/r03h_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03h_tests.log; tail -4 gpurun_out/r03h_tests.log
cd /tmp && export TMPDIR=/tmp
for x in 0 1; do
export RSA_RING_XRES=$x
rm -rf /tmp/prof_s
rocprofv3 --kernel-trace --stats -d /tmp/prof_s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-power > $GRAFT_REPO_ROOT/gpurun_out/r03h_prof_x$x.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_export.py stats $(find /tmp/prof_s -name '*.db' | head -1) $GRAFT_REPO_ROOT/gpurun_out/r03h_kernel_stats_x$x.csv
head -8 $GRAFT_REPO_ROOT/gpurun_out/r03h_kernel_stats_x$x.csv | cut -c1-150
done
