#!/bin/bash
# runs on the GPU box: SQ counters + clock per kernel family of `bench.py ARGS` (two --pmc passes, no other tracing) -> gpurun_out/<tag>_sq_counters.txt
# usage: tools/prof_sq.sh TAG [bench.py args, e.g. --config c4]
set -e
R=$GRAFT_REPO_ROOT
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sq1 /tmp/sq2
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/sq1 -- python3 $R/bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-roofline --no-power > $R/gpurun_out/${tag}_sq1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/sq2 -- python3 $R/bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-roofline --no-power > $R/gpurun_out/${tag}_sq2.log 2>&1
python3 $R/tools/sq_summary.py $(find /tmp/sq1 -name '*.db' | head -1) $(find /tmp/sq2 -name '*.db' | head -1) $R/gpurun_out/${tag}_sq_counters.txt > /dev/null
cut -c1-420 $R/gpurun_out/${tag}_sq_counters.txt
