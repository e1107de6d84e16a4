#!/bin/bash
# Experiment helper (not part of the product build): rebuild the ring-schedule translation units with extra flags and link them
# with the product objects of build/obj into variants/lib_NAME.so.
# usage: tools/variant.sh NAME "-DFLAG=.. ..." [file.hip ...]      (default files: conv_inst_ring1 conv_inst_ring2)
set -e
name=$1; flags=$2; shift 2 || true
files=${@:-conv_inst_ring1 conv_inst_ring2}
root=$(cd "$(dirname "$0")/.." && pwd)
od=$root/build/obj_$name; mkdir -p $od
pids=()
for f in $files; do
  slp=""; case $f in conv_inst_ring*) slp="-fno-slp-vectorize";; esac   # as resselt_amd/build.py compiles these units
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $slp -std=c++17 -fPIC -c $flags -I$root/include -I$root/resselt_amd/csrc $root/resselt_amd/csrc/$f.hip -o $od/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
objs=""
for o in $root/build/obj/*.o; do b=$(basename $o); if [ -f $od/$b ]; then objs="$objs $od/$b"; else objs="$objs $o"; fi; done
mkdir -p $root/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $root/variants/lib_$name.so
echo built variants/lib_$name.so
