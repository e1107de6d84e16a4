cd $GRAFT_REPO_ROOT
for i in 1 2; do python tools/conv5_time.py 2>&1 | tail -1; for v in infl1 infl3; do RSA_LIB=variants/lib_$v.so python tools/conv5_time.py 2>&1 | tail -1; done; done | tee gpurun_out/r03k_conv5_infl.log
for v in "" infl1 infl3; do if [ -n "$v" ]; then export RSA_LIB=variants/lib_$v.so; fi; timeout -k 10 100 python bench.py --config c3 --no-cpu-baseline --no-kernel-roofline --no-power 2>/dev/null | cut -c1-250 | sed "s/^/$v /" | tee -a gpurun_out/r03k_c3_infl.log; done
