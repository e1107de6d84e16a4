cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_conv_fp16_gpu.py tests/test_rrdbnet_gpu.py -x -q -m gpu > gpurun_out/r03j_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03j_tests.log; tail -3 gpurun_out/r03j_tests.log
for i in 1 2; do RSA_RING_XRES=0 python tools/conv5_time.py 2>&1 | tail -1; python tools/conv5_time.py 2>&1 | tail -1; for pd in 0 2; do RSA_LIB=variants/lib_pdx$pd.so python tools/conv5_time.py 2>&1 | tail -1; done; done | tee gpurun_out/r03j_conv5_pdx.log
for i in 1 2; do
RSA_RING_XRES=0 timeout -k 10 200 python tools/frame_time.py auto 6 2>&1 | tail -1 | sed 's/^/xres0 /' | tee -a gpurun_out/r03j_xres.log
RSA_RING_XRES=1 timeout -k 10 200 python tools/frame_time.py auto 6 2>&1 | tail -1 | sed 's/^/xres1 /' | tee -a gpurun_out/r03j_xres.log
done
