#!/bin/bash
cd /tmp && export TMPDIR=/tmp
i=0
for v in "SPMV_TILED_LANE_ENTRIES=4 SPMV_TILED_REDUCE_BLOCK=1024" "SPMV_TILED_LANE_ENTRIES=2 SPMV_TILED_REDUCE_BLOCK=1024" "SPMV_TILED_LANE_ENTRIES=4 SPMV_TILED_REDUCE_BLOCK=512"; do
  i=$((i+1)); d=$GRAFT_REPO_ROOT/gpurun_out/p2f64_$i
  env $v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python3 $GRAFT_REPO_ROOT/tools/quick_bench.py c5only > $d.log 2>&1
  echo "== $v"; grep "c5 10M" $d.log; python3 $GRAFT_REPO_ROOT/tools/kstats.py $d/r_kernel_stats.csv | grep "tiled_"
done
