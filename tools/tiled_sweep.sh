#!/bin/bash
# developer probe: tiled-engine shape sweep on the GPU box (C5)
for cfg in "8192 8192 2" "16384 8192 4" "16384 8192 2" "16384 4096 2" "32768 4096 4" "32768 8192 4" "32768 2048 2"; do
  set -- $cfg
  echo "== strip $1 tile $2 chunks $3"
  SPMV_TILED_STRIP=$1 SPMV_TILED_TILE=$2 SPMV_TILED_CHUNKS=$3 timeout -k 5 120 python tools/quick_bench.py c5only 2>&1 | grep "kernel=11"
done
