#!/bin/bash
# kernel-level A/B of the two forms of phase 2 on one box: rocprofv3 --kernel-trace --stats over tools/quick_bench.py c5only c2only,
# twice each, interleaved
cd "$(dirname "$0")/.."
out=gpurun_out/stream_ab.txt
mkdir -p gpurun_out; : > $out
for round in 1 2; do
  for form in 1 0; do
    d=gpurun_out/stream_ab_${form}_$round
    ( cd /tmp && export TMPDIR=/tmp && SPMV_TILED_STREAM=$form timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$d -o r -- python3 $GRAFT_REPO_ROOT/tools/quick_bench.py c5only c2only ) > $d.log 2>&1 || exit 1
    echo "== SPMV_TILED_STREAM=$form (round $round)" >> $out
    grep "kernel=" $d.log >> $out
    python3 tools/kstats.py $d/r_kernel_stats.csv | grep "tiled_" >> $out
  done
done
cat $out
