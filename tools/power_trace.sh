#!/bin/bash
# runs on the GPU box: samples socket power / clocks (rocm-smi) while a command runs in the background -> gpurun_out/$1
# usage: tools/power_trace.sh OUT.txt python3 bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-kernel-roofline
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-power_trace.txt}; shift
{
  echo "# idle:"; rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -E "Power|sclk|mclk|Max" | head -8
  "$@" > $out.cmd.log 2>&1 &
  pid=$!
  echo "# while '$*' runs (one sample per rocm-smi call):"
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" | tr '\n' ' ' | sed 's/=\+//g; s/  */ /g'
    echo
  done
  wait $pid
  echo "# command exit $?"; tail -c 400 $out.cmd.log
} > $out 2>&1
