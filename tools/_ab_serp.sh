cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_conv_fp16_gpu.py -x -q -m gpu -k "reversed or one_fp16" > gpurun_out/r03c_test.log 2>&1; echo "rc=$?" >> gpurun_out/r03c_test.log; tail -3 gpurun_out/r03c_test.log
for i in 1 2; do
RSA_SERPENTINE=0 timeout -k 10 200 python tools/frame_time.py auto 6 2>&1 | tail -1 | sed 's/^/serp0 /' | tee -a gpurun_out/r03c_serp.log
RSA_SERPENTINE=1 timeout -k 10 200 python tools/frame_time.py auto 6 2>&1 | tail -1 | sed 's/^/serp1 /' | tee -a gpurun_out/r03c_serp.log
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -- python3 $GRAFT_REPO_ROOT/bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-power > $GRAFT_REPO_ROOT/gpurun_out/r03c_c4_prof.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_export.py stats $(find /tmp/prof_c4 -name '*.db' | head -1) $GRAFT_REPO_ROOT/gpurun_out/r03c_kernel_stats_c4.csv
head -12 $GRAFT_REPO_ROOT/gpurun_out/r03c_kernel_stats_c4.csv | cut -c1-200
