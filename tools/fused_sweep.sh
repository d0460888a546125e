#!/bin/bash
# sweeps of tools/fused_bench on the GPU box; every run bounded by its own timeout
cd "$(dirname "$0")/.."
out=gpurun_out/fused_bench.txt
mkdir -p gpurun_out
: > $out
run() { echo "== $*" >> $out; timeout -k 10 90 tools/fused_bench "$@" >> $out 2>&1 || { echo "FAILED rc=$? : $*" >> $out; return 1; }; }
# W R D H P E reps mode batch — and the two-kernel engine on the same box, before and after
python tools/quick_bench.py c5only 2>&1 | grep kernel= >> $out
run 16384 9792 2 1 3 4 5 0 16 &&
run 16384 9792 3 1 3 4 5 0 16 &&
run 16384 9792 4 1 3 4 5 0 16 &&
run 16384 9792 3 1 3 4 5 0 64 &&
run 16384 9792 4 2 3 4 5 0 16
python tools/quick_bench.py c5only 2>&1 | grep kernel= >> $out
grep -v "^rows diff\|^registration" $out
