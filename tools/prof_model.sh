#!/bin/bash
# runs on the GPU box: rocprofv3 kernel stats of one secondary config (tools/profile_model.py NAME) -> gpurun_out/TAG_kernel_stats_NAME.csv
set -e
R=$GRAFT_REPO_ROOT
name=${1:-swinir}; tag=${2:-r02}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pm_$name -- python3 $R/tools/profile_model.py $name > $R/gpurun_out/${tag}_pm_$name.log 2>&1
python3 $R/tools/rocpd_export.py stats $(find /tmp/pm_$name -name '*.db' | head -1) $R/gpurun_out/${tag}_kernel_stats_$name.csv
head -12 $R/gpurun_out/${tag}_kernel_stats_$name.csv | cut -c1-200
