#!/bin/bash
# developer probe: strip width / tile height sweep of the tiled engine on C5 (drop-in spmv_csr)
out=gpurun_out/shape_sweep.txt
: > $out
run() { echo "== $*" >> $out; env "$@" python tools/quick_bench.py c5only 2>&1 | grep "kernel=" >> $out; }
run SPMV_TILED_STRIP=16384 SPMV_TILED_TILE=9792
run SPMV_TILED_STRIP=32768 SPMV_TILED_TILE=9792
run SPMV_TILED_STRIP=8192 SPMV_TILED_TILE=9792
run SPMV_TILED_STRIP=16384 SPMV_TILED_TILE=6528
run SPMV_TILED_STRIP=32768 SPMV_TILED_TILE=6528
run SPMV_TILED_STRIP=16384 SPMV_TILED_TILE=4928
run SPMV_TILED_STRIP=32768 SPMV_TILED_TILE=4928
cat $out
