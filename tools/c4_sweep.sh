#!/bin/bash
# developer probe: strip width / phase-1 item size / tile rows on the 1 M power-law matrix (C4)
out=gpurun_out/c4_sweep.txt
: > $out
run() { echo "== $*" >> $out; env "$@" python tools/quick_bench.py c4only 2>&1 | grep "kernel=12" >> $out; }
run SPMV_TILED_FOLD=0
run SPMV_DEBUG=item=8192 SPMV_TILED_FOLD=0
run SPMV_DEBUG=item=32768 SPMV_TILED_FOLD=0
run SPMV_DEBUG=strip=8192 SPMV_TILED_FOLD=0
run SPMV_DEBUG=strip=8192,item=8192 SPMV_TILED_FOLD=0
run SPMV_DEBUG=strip=8192,item=16384 SPMV_TILED_FOLD=0
run SPMV_DEBUG=strip=4096,item=8192 SPMV_TILED_FOLD=0
run SPMV_DEBUG=strip=32768,item=32768 SPMV_TILED_FOLD=0
run SPMV_DEBUG=tile=1024 SPMV_TILED_FOLD=0
run SPMV_DEBUG=tile=4096 SPMV_TILED_FOLD=0
cat $out
