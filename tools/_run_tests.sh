cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_stream_kernels_gpu.py tests/test_rrdbnet_gpu.py tests/test_baseline_configs_gpu.py -x -q -m gpu -s > gpurun_out/t4.log 2>&1; echo "rc=$?" >> gpurun_out/t4.log; grep -v "^$" gpurun_out/t4.log | tail -25
