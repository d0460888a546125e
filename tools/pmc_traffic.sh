#!/bin/bash
# HBM traffic of the step kernels: separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE
# (MI355X_MICROARCH.md, HBM section), general path and value-folded path.  Run on the GPU box from
# the repo root; tools/pmc_traffic.py turns the CSVs into profiles/pmc_traffic.json.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic
mkdir -p $OUT
run() {
  name=$1; counter=$2
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $counter --output-format csv -d $OUT/$name -o $name -- \
      python3 $GRAFT_REPO_ROOT/bench.py --no-extras --steps 5 --warmup 1 > $OUT/$name.log 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
}
export SPMV_TILED_FOLD=0
run general_fetch FETCH_SIZE
run general_write WRITE_SIZE
export SPMV_TILED_FOLD=1
run folded_fetch FETCH_SIZE
run folded_write WRITE_SIZE
