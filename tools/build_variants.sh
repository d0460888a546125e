#!/bin/bash
# builds variants of the library that differ in one -D of csrc/tiled.hip into tools/probe_libs/ (git-ignored, travels
# with gpurun; loaded through SPMV_AMD_LIB).  usage: tools/build_variants.sh NAME=-DFLAG[=V] ...
set -e
cd "$(dirname "$0")/../gpu-spmv_amd"
make -s -j8
mkdir -p ../tools/probe_libs
objs=$(ls build/*.o | grep -v '/tiled.o$' | grep -v '/tiled_var_')
for spec in "$@"; do
  name=${spec%%=*}; flag=${spec#*=}
  /opt/rocm/bin/hipcc -x hip -std=c++17 -O3 -fPIC -I../include -Icsrc -Wall -Wno-unused-function --offload-arch=gfx950 \
      -ffp-contract=on $flag -c csrc/tiled.hip -o build/tiled_var_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build/tiled_var_$name.o -o ../tools/probe_libs/libspmv_$name.so
  echo "built tools/probe_libs/libspmv_$name.so ($flag)"
done
