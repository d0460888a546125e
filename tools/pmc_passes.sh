#!/bin/bash
# developer probe: stall / occupancy counters of the step kernels, one rocprofv3 --pmc pass per group
# (run on the GPU box from the repo root).  Stops at the first pass that times out or is killed.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc2
mkdir -p $OUT
run() {
  name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- \
      python3 $GRAFT_REPO_ROOT/bench.py --no-extras --steps 5 --warmup 2 > $OUT/$name.log 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
run sq2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum
run drv MemUnitStalled MeanOccupancyPerCU LdsBankConflict
