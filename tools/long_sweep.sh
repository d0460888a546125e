#!/bin/bash
# developer probe: where should a row stop being "long" (direct path) now that phase 2 adds with ds_add_f64?
out=gpurun_out/long_sweep.txt
: > $out
run() { echo "== $*" >> $out; env "$@" python tools/quick_bench.py c4only c5pl 2>&1 | grep "kernel=12\|plan" >> $out; }
run SPMV_TILED_LONG_FACTOR=2
run SPMV_TILED_LONG_FACTOR=4
run SPMV_TILED_LONG_FACTOR=8 SPMV_TILED_LONG_CAP=4096
run SPMV_TILED_LONG_FACTOR=16 SPMV_TILED_LONG_CAP=8192
run SPMV_TILED_LONG_FACTOR=1000 SPMV_TILED_LONG_CAP=100000
cat $out
