#!/bin/bash
# developer probe: fabric read-request counters of the tiled kernels on C5 (drop-in spmv_csr path)
# usage (GPU box, repo root): tools/pmc_quick.sh <tag> [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcq_$tag
mkdir -p $OUT
run() {
  name=$1; shift
  env "${EXTRA[@]}" timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- \
      python3 $GRAFT_REPO_ROOT/tools/quick_bench.py c5only > $OUT/$name.log 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
}
EXTRA=("$@")
[ ${#EXTRA[@]} -eq 0 ] && EXTRA=(SPMV_DUMMY=1)
if [ "$PMC_SET" = "sq" ]; then
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES
else
run tcc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run hit TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
fi
python3 $GRAFT_REPO_ROOT/tools/pmc_quick.py $OUT
