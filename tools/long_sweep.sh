#!/bin/bash
# developer probe: where should a row stop being "long" (direct path) now that phase 2 adds with ds_add_f64?
out=gpurun_out/long_sweep.txt
: > $out
run() { echo "== $*" >> $out; env "$@" python tools/quick_bench.py c4only c5pl 2>&1 | grep "kernel=12\|plan" >> $out; }
run SPMV_DEBUG=long_factor=2
run SPMV_DEBUG=long_factor=4
run SPMV_DEBUG=long_factor=8,long_cap=4096
run SPMV_DEBUG=long_factor=16,long_cap=8192
run SPMV_DEBUG=long_factor=1000,long_cap=100000
cat $out
