cd $GRAFT_REPO_ROOT
RSA_LIB=variants/lib_dbg.so MODE=fp16 timeout -k 10 300 python tools/ring_ablate.py 160,32 64,32 192,64 -- 0 30 14 6 22 1 2>&1 | tail -3 | tee gpurun_out/r03n_ring_abl_fill.log
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/r03n_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r03n_tests.log; tail -3 gpurun_out/r03n_tests.log
