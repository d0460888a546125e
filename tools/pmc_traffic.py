"""Turns the CSVs of tools/pmc_traffic.sh (gpurun_out/pmc_traffic/) into profiles/pmc_traffic.json."""
import collections
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "pmc_traffic")


def per_kernel(name):
    acc = collections.defaultdict(list)
    path = os.path.join(SRC, name, name + "_counter_collection.csv")
    for row in csv.DictReader(open(path)):
        kernel = row["Kernel_Name"]
        for key in ("tiled_expand_kernel", "tiled_pagerank_reduce_kernel"):
            if key in kernel:
                acc[key].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    out = {
        "command": "tools/pmc_traffic.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) "
                   "-- python3 bench.py --no-extras --steps 5 --warmup 1, with SPMV_TILED_FOLD=0 and =1",
        "correction": "gfx950: FETCH_SIZE counts 1/2 of the bytes of coalesced streaming reads (MI355X_MICROARCH.md HBM "
                      "section; re-checked in round 1 on count_columns_kernel: 312 MB reported for a 640 MB "
                      "dword-per-lane stream), so read bytes = 2 x FETCH_SIZE; WRITE_SIZE taken as is; both in KB",
        "algorithmic_bytes_per_step": 1400000004,
    }
    for variant in ("general", "folded"):
        fetch, write = per_kernel(variant + "_fetch"), per_kernel(variant + "_write")
        kernels, total = {}, 0.0
        for k in fetch:
            kernels[k] = {"FETCH_SIZE_KB": fetch[k], "WRITE_SIZE_KB": write[k],
                          "bytes": (2 * fetch[k] + write[k]) * 1024}
            total += kernels[k]["bytes"]
        out[variant] = {"kernels_avg_per_launch": kernels, "bytes_per_step": int(total)}
    out["pr_step_kernel_bytes_per_launch"] = out["general"]["bytes_per_step"]     # what bench.py reports as roofline.traffic
    # the plan shape the passes ran on (bench.py reports the traffic only for the same shape) and when
    for line in open(os.path.join(SRC, "general_fetch.log")):
        if line.startswith("{"):
            plan = json.loads(line)["roofline"]["tiled_plan"]
            out["plan"] = {k: plan[k] for k in ("strip_cols", "tile_rows", "num_strips", "num_tiles", "slots_in_cells")}
    # which build the passes ran on: bench.py reports the traffic only while the kernels' sources are the ones measured
    import hashlib
    out["sources_sha256"] = {name: hashlib.sha256(open(os.path.join(ROOT, "gpu-spmv_amd", "csrc", name), "rb").read()).hexdigest()
                             for name in ("tiled.hip", "pagerank.hip")}
    out["commit"] = os.environ.get("SPMV_COMMIT", "unknown (set SPMV_COMMIT=$(git rev-parse --short HEAD) in the gpurun command)")
    import datetime
    out["collected"] = "tools/pmc_traffic.sh, separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, %s" % (
        datetime.date.today().isoformat())
    out["note"] = ("sum over the two launches of one PageRank step (tiled_expand_kernel + tiled_pagerank_reduce_kernel); "
                   "bench.py's headline runs the general path")
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps({v: out[v]["bytes_per_step"] for v in ("general", "folded")}))


if __name__ == "__main__":
    main()
