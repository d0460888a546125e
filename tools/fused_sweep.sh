#!/bin/bash
# sweeps of tools/fused_bench on the GPU box; every run bounded by its own timeout
cd "$(dirname "$0")/.."
out=gpurun_out/fused_bench.txt
mkdir -p gpurun_out
: > $out
run() { echo "== $*" >> $out; timeout -k 10 90 tools/fused_bench "$@" >> $out 2>&1 || { echo "FAILED rc=$? : $*" >> $out; return 1; }; }
# <local|global|split> W R D H PW|P reps mode
run split 16384 9792 2 1 2 5 0 &&
run split 16384 9792 2 1 2 5 1 &&
run split 16384 9792 2 1 2 5 2 &&
run split 16384 9792 2 1 4 5 0 &&
run split 16384 9792 2 1 4 5 2 &&
run split 16384 9792 2 2 2 5 0 &&
run split 16384 9792 3 2 2 5 0 &&
run split 16384 9792 3 4 2 5 0
grep -v "^rows diff" $out
