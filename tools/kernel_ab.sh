#!/bin/bash
# kernel-level A/B of library variants on one box: rocprofv3 --kernel-trace --stats over tools/quick_bench.py c5only, every
# variant twice, interleaved; prints the tiled kernels' average durations.
# usage: tools/kernel_ab.sh "" tools/probe_libs/libspmv_x.so "SPMV_DEBUG=parts=4 tools/probe_libs/libspmv_y.so" ...
#        (an empty string = the product library; words with '=' in front of the library are environment assignments)
cd "$(dirname "$0")/.."
out=gpurun_out/kernel_ab.txt
mkdir -p gpurun_out; : > $out
WHICH=${WHICH:-c5only}
for round in $(seq 1 ${ROUNDS:-2}); do
  i=0
  for spec in "$@"; do
    i=$((i + 1))
    d=gpurun_out/kernel_ab_${i}_$round
    lib=""; envs=""
    for word in $spec; do case "$word" in *=*) envs="$envs $word";; *) lib=$word;; esac; done
    ( cd /tmp && export TMPDIR=/tmp $envs && SPMV_AMD_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$d -o r -- python3 $GRAFT_REPO_ROOT/tools/quick_bench.py $WHICH ) > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
    echo "== variant $i [${spec:-product}] round $round" >> $out
    grep "kernel=" $d.log >> $out
    python3 tools/kstats.py $d/r_kernel_stats.csv | grep "tiled_" >> $out
  done
done
cat $out
