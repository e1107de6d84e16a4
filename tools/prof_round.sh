#!/bin/bash
# runs on the GPU box: rocprofv3 kernel stats + SQ counters + FETCH/WRITE passes of the headline bench -> gpurun_out/<tag>_*
# usage: tools/prof_round.sh TAG      (copy the files you want judged from gpurun_out/ into profiles/)
set -e
R=$GRAFT_REPO_ROOT
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_s -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-power --no-secondary > $R/gpurun_out/${tag}_prof_s.log 2>&1
python3 $R/tools/rocpd_export.py stats $(find /tmp/prof_s -name '*.db' | head -1) $R/gpurun_out/${tag}_kernel_stats_bench.csv
echo stats done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/sq1 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-roofline --no-power --no-secondary > $R/gpurun_out/${tag}_sq1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/sq2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-roofline --no-power --no-secondary > $R/gpurun_out/${tag}_sq2.log 2>&1
python3 $R/tools/sq_summary.py $(find /tmp/sq1 -name '*.db' | head -1) $(find /tmp/sq2 -name '*.db' | head -1) $R/gpurun_out/${tag}_sq_counters.txt > /dev/null
echo sq done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_f -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-roofline --no-power --no-secondary > $R/gpurun_out/${tag}_prof_f.log 2>&1
python3 $R/tools/rocpd_export.py pmc $(find /tmp/prof_f -name '*.db' | head -1) /tmp/fetch.csv
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_w -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-roofline --no-power --no-secondary > $R/gpurun_out/${tag}_prof_w.log 2>&1
python3 $R/tools/rocpd_export.py pmc $(find /tmp/prof_w -name '*.db' | head -1) /tmp/write.csv
# forwards in a `--steps 1 --warmup 0` bench run: the timed step + one untimed and one timed forward of the kernel-time leg
python3 $R/tools/hbm_traffic.py /tmp/fetch.csv /tmp/write.csv 3 $R/gpurun_out/${tag}_hbm_traffic.json > /dev/null
echo traffic done
cut -c1-330 $R/gpurun_out/${tag}_sq_counters.txt
head -c 1200 $R/gpurun_out/${tag}_hbm_traffic.json
head -8 $R/gpurun_out/${tag}_kernel_stats_bench.csv
