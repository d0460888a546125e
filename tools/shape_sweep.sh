#!/bin/bash
# strip / tile shape sweep of the tiled engine on C5 (SPMV_DEBUG=strip=W,tile=R), one box
cd "$(dirname "$0")/.."
out=gpurun_out/shape_sweep.txt
mkdir -p gpurun_out; : > $out
run() { echo "== $*" >> $out; env "$@" python tools/quick_bench.py c5only 2>&1 | grep "kernel=" >> $out; }
for strip in 16384 32768; do
  for tile in 9792 6528 4928 3264; do
    run SPMV_DEBUG=strip=$strip,tile=$tile
  done
done
run SPMV_DEBUG=strip=16384,tile=9792
run SPMV_DEBUG=strip=8192,tile=9792
cat $out
