#!/bin/bash
# the per-round evidence in one call (run on the GPU box from the repo root): kernel stats of `bench.py --no-extras` under
# rocprofv3, the four PMC traffic passes (-> profiles/pmc_traffic.json with the sources' hashes and SPMV_COMMIT), the full bench line.
# usage: SPMV_COMMIT=<hash> tools/collect_profiles.sh r04
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/${TAG}_stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -o $TAG -- python3 $R/bench.py --no-extras > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_bench_under_rocprof.err || exit 1
echo "stats done"
bash $R/tools/pmc_traffic.sh || exit 1
cd $R && python3 tools/pmc_traffic.py && cp profiles/pmc_traffic.json gpurun_out/pmc_traffic_${TAG}.json
python3 bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_line.err; echo "bench rc=$?"
