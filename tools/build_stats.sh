#!/bin/bash
# kernel times of the plan build (C2 + C5 builds of tools/build_time.py) under rocprofv3 --stats
cd "$(dirname "$0")/.."
d=gpurun_out/build_stats${1:+_$1}
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$d -o r -- python3 $GRAFT_REPO_ROOT/tools/build_time.py ) > $d.log 2>&1 || exit 1
grep "first call" $d.log
python3 tools/kstats.py $d/r_kernel_stats.csv | grep -v "uniform_rows\|vector_kernel\|tiled_expand\|tiled_reduce\|copyBuffer\|fillBuffer"
