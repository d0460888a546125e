cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03_stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_stats -o r03 -- python3 $R/bench.py --no-extras > $R/gpurun_out/r03_bench_under_rocprof.json 2> $R/gpurun_out/r03_bench_under_rocprof.err || exit 1
echo "stats done"
bash $R/tools/pmc_traffic.sh || exit 1
cd $R && python3 tools/pmc_traffic.py && cp profiles/pmc_traffic.json gpurun_out/pmc_traffic_r03.json
python3 bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.err; echo "bench rc=$?"
