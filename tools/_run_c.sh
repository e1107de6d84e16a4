cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py tests/test_rrdbnet_gpu.py tests/test_swinir_gpu.py -x -q -m gpu > gpurun_out/r03e_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03e_tests.log; tail -3 gpurun_out/r03e_tests.log
timeout -k 10 200 python tools/frame_time.py auto 6 2>&1 | tail -1 | tee -a gpurun_out/r03e_frame.log
timeout -k 10 200 python tools/dbg/c4_time.py auto 2>&1 | tail -1 | tee -a gpurun_out/r03e_sb_abl.log
for m in 4 8 12 32 2 1 16; do RSA_LIB=variants/lib_sb$m.so timeout -k 10 200 python tools/dbg/c4_time.py auto 2>&1 | tail -1 | tee -a gpurun_out/r03e_sb_abl.log; done
RSA_LIB=variants/lib_dbg.so MODE=fp16 timeout -k 10 300 python tools/ring_ablate.py 64,32 160,32 192,64 2>&1 | tail -4 | tee -a gpurun_out/r03e_ring_abl.log
