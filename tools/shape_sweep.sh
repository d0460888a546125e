#!/bin/bash
# strip / tile shape sweep of the tiled engine on C5 (env knobs SPMV_TILED_STRIP / SPMV_TILED_TILE), one box
cd "$(dirname "$0")/.."
out=gpurun_out/shape_sweep.txt
mkdir -p gpurun_out; : > $out
run() { echo "== $*" >> $out; env "$@" python tools/quick_bench.py c5only 2>&1 | grep "kernel=" >> $out; }
for strip in 16384 32768; do
  for tile in 9792 6528 4928 3264; do
    run SPMV_TILED_STRIP=$strip SPMV_TILED_TILE=$tile
  done
done
run SPMV_TILED_STRIP=16384 SPMV_TILED_TILE=9792
run SPMV_TILED_STRIP=16384 SPMV_TILED_TILE=9792 SPMV_TILED_STREAM=0
run SPMV_TILED_STRIP=8192 SPMV_TILED_TILE=9792
cat $out
