#!/bin/bash
# developer probe: SQ / TCC counters of the plan-build kernels (C2 + C5 builds of tools/build_time.py)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_build
mkdir -p $OUT
run() {
  name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- \
      python3 $GRAFT_REPO_ROOT/tools/build_time.py > $OUT/$name.log 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
}
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES
run tcc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum
python3 - <<'PY'
import collections, csv, glob, os, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_build")
for path in glob.glob(os.path.join(out, "*", "*_counter_collection.csv")):
    for row in csv.DictReader(open(path)):
        k = re.sub(r"\(anonymous namespace\)::|spmv::detail::|void ", "", row["Kernel_Name"]).split("(")[0]
        if "batch_" in k or "cell_place" in k or "max_row" in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print("   %-28s" % c, " ".join("%14.0f" % v for v in acc[k][c]))
PY
