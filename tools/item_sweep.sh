#!/bin/bash
# developer probe: phase-1 work item size (slots per workgroup per x-strip load) on C5
cd /tmp && export TMPDIR=/tmp
for item in 16384 32768 49152 65536; do
  d=$GRAFT_REPO_ROOT/gpurun_out/item_$item
  SPMV_DEBUG=item=$item timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python3 $GRAFT_REPO_ROOT/tools/quick_bench.py c5only > $d.log 2>&1
  echo "== item=$item"; grep "kernel=" $d.log; python3 $GRAFT_REPO_ROOT/tools/kstats.py $d/r_kernel_stats.csv | grep "tiled_"
done
