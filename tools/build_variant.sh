#!/bin/bash
# A/B builds of one kernel file: tools/build_variant.sh NAME SOURCE "EXTRA FLAGS" -> build/ab/NAME.so (the other objects from build/obj).
# Run on the CPU box after `python __graft_entry__.py`; select with RPT_LIB=build/ab/NAME.so.
set -e
name=$1; src=$2; extra=$3
base=$(basename "${src%.*}")
mkdir -p build/ab
[ "$src" = "kernels_f64.hip" ] && extra="-mllvm -disable-machine-licm $extra"   # (__graft_entry__.FILE_FLAGS)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC $extra -c rpt_amd/csrc/$src -o build/ab/$name.$base.o
objs=""
for o in build/obj/*.o; do
  if [ "$(basename $o)" = "$base.o" ]; then objs="$objs build/ab/$name.$base.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -Wl,-rpath,/opt/rocm/lib -o build/ab/$name.so $objs -ldl
echo built build/ab/$name.so
