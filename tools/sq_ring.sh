#!/bin/bash
# runs on the GPU box: SQ counters + clock of single-layer launches (tools/conv_microbench.py) -> gpurun_out/$1
set -e
R=$GRAFT_REPO_ROOT
out=${1:-sq_ring.txt}; shift || true
cd /tmp && export TMPDIR=/tmp
export MB_ITERS=20
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/sqa -- python3 $R/tools/conv_microbench.py "$@" > $R/gpurun_out/sqa.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/sqb -- python3 $R/tools/conv_microbench.py "$@" > $R/gpurun_out/sqb.log 2>&1
python3 $R/tools/sq_summary.py $(find /tmp/sqa -name '*.db' | head -1) $(find /tmp/sqb -name '*.db' | head -1) $R/gpurun_out/$out
