#!/bin/bash
cd /tmp && export TMPDIR=/tmp
i=0
for v in "SPMV_TILED_LANE_ENTRIES=8" "SPMV_TILED_LANE_ENTRIES=8 SPMV_TILED_P2_WIDE=1" "SPMV_TILED_LANE_ENTRIES=4"; do
  i=$((i+1)); d=$GRAFT_REPO_ROOT/gpurun_out/p2e8_$i
  env $v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python3 $GRAFT_REPO_ROOT/tools/quick_bench.py c5only c2only c4only > $d.log 2>&1
  echo "== $v"; grep "kernel=" $d.log; python3 $GRAFT_REPO_ROOT/tools/kstats.py $d/r_kernel_stats.csv | grep "tiled_"
done
