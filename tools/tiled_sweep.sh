#!/bin/bash
# developer probe: tiled-engine shape sweep on the GPU box (C5)
for cfg in "8192 8192 512" "8192 8192 1024" "8192 4096 1024" "16384 4096 1024" "16384 8192 1024" "8192 2048 1024"; do
  set -- $cfg
  echo "== strip $1 tile $2 block $3"
  SPMV_TILED_STRIP=$1 SPMV_TILED_TILE=$2 SPMV_TILED_RBLOCK=$3 timeout -k 5 120 python tools/quick_bench.py c5only 2>&1 | grep "kernel=11"
done
