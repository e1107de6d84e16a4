cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_conv_fp16_gpu.py tests/test_conv_gpu.py tests/test_rrdbnet_gpu.py -x -q -m gpu > gpurun_out/r03h_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03h_tests.log; tail -4 gpurun_out/r03h_tests.log
for i in 1 2; do
RSA_RING_XRES=0 timeout -k 10 200 python tools/frame_time.py auto 6 2>&1 | tail -1 | sed 's/^/xres0 /' | tee -a gpurun_out/r03h_xres.log
RSA_RING_XRES=1 timeout -k 10 200 python tools/frame_time.py auto 6 2>&1 | tail -1 | sed 's/^/xres1 /' | tee -a gpurun_out/r03h_xres.log
done
