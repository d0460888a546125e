#!/bin/bash
# developer probe: tiled-engine timings (drop-in spmv_csr, use_texture) + plan build time
out=gpurun_out/v2_sweep.txt
: > $out
run() { echo "== $*" >> $out; env "$@" python tools/quick_bench.py ${WHICH:-c5only} 2>&1 | grep -v "^spmv-amd\|amdgpu.ids" >> $out; }
WHICH="c5only c2only c4only" run SPMV_DUMMY=1
echo "== build time" >> $out
python tools/build_time.py 2>&1 | grep -v amdgpu.ids >> $out
cat $out
