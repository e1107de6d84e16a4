cd $GRAFT_REPO_ROOT
RSA_RING_PREFETCH=3 timeout -k 10 120 python -m pytest tests/test_conv_fp16_gpu.py -x -q -m gpu -k "one_fp16_product or conv5" 2>&1 | tail -2 && \
RSA_RING_PREFETCH=3 timeout -k 10 300 python -m pytest tests/test_conv_fp16_gpu.py tests/test_rrdbnet_gpu.py -x -q -m gpu 2>&1 | tail -2 && \
for pf in 0 1 2 3 4 6 8 0; do RSA_RING_PREFETCH=$pf timeout -k 10 200 python tools/frame_time.py auto 5 2>&1 | tail -1 | sed "s/^/pf$pf /" | tee -a gpurun_out/r03o_prefetch.log; done
