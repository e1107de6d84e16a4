#!/bin/bash
# runs on the GPU box: SQ counters + clock of the fused Swin block kernels alone (tools/swin_block_bench.py) -> gpurun_out/$1
set -e
R=$GRAFT_REPO_ROOT
out=${1:-sq_swin.txt}; shift || true
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/sqsa -- python3 $R/tools/swin_block_bench.py "$@" > $R/gpurun_out/sqsa.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/sqsb -- python3 $R/tools/swin_block_bench.py "$@" > $R/gpurun_out/sqsb.log 2>&1
python3 $R/tools/sq_summary.py $(find /tmp/sqsa -name '*.db' | head -1) $(find /tmp/sqsb -name '*.db' | head -1) $R/gpurun_out/$out
