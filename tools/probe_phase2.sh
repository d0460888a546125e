#!/bin/bash
# where does phase 2's time go?  the shipped library against two timing-only builds of csrc/tiled.hip
# (-DSPMV_PROBE_PHASE2=1: plain LDS stores where the ds_add_f64 go; =2: no LDS traffic for the adds; =3: a tile's runs read back to back);
# build them first, in the build container: make -C gpu-spmv_amd probes.  Results of the probes are wrong by construction.  SPMV_TILED_STREAM=0:
# the probes sit in the run-by-run form of phase 2
cd "$(dirname "$0")/.."
out=gpurun_out/probe_phase2.txt
mkdir -p gpurun_out; : > $out
export SPMV_TILED_STREAM=0
for lib in "" tools/probe_libs/libspmv_probe1.so tools/probe_libs/libspmv_probe2.so tools/probe_libs/libspmv_probe3.so; do
  echo "== lib ${lib:-shipped}" >> $out
  d=gpurun_out/probe_$(basename "${lib:-shipped}" .so)
  ( cd /tmp && export TMPDIR=/tmp && SPMV_AMD_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$d -o r -- python3 $GRAFT_REPO_ROOT/tools/quick_bench.py c5only ) >> $out 2>&1 || exit 1
  python3 tools/kstats.py $d/r_kernel_stats.csv | grep "tiled_" >> $out
done
cat $out | grep -v "^W\|amdgpu.ids"
