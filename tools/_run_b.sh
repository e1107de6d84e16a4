cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r03d_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03d_tests.log; tail -3 gpurun_out/r03d_tests.log
for c in c3 c4; do
RSA_SERPENTINE=0 timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --no-kernel-roofline --no-power 2>/dev/null | cut -c1-330 | sed 's/^/serp0 /' | tee -a gpurun_out/r03d_serp_$c.log
RSA_SERPENTINE=1 timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --no-kernel-roofline --no-power 2>/dev/null | cut -c1-330 | sed 's/^/serp1 /' | tee -a gpurun_out/r03d_serp_$c.log
done
bash tools/prof_sq.sh r03d_c4 --config c4
