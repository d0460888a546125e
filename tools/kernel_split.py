"""kernel_split.py — per-matrix averages of the tiled engine's two kernels from the rocprofv3 kernel traces tools/kernel_ab.sh
leaves (gpurun_out/kernel_ab_<variant>_<round>/r_kernel_trace.csv): quick_bench.py times every matrix with 13 calls, so the
calls of each kernel are cut into consecutive groups of 13, in launch order.  usage: python tools/kernel_split.py [dir ...]"""
import collections
import csv
import glob
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def split(path, per_matrix=13):
    calls = collections.defaultdict(list)
    for r in sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"])):
        name = r["Kernel_Name"]
        for key in ("tiled_expand", "tiled_reduce", "tiled_pagerank_reduce"):
            if key + "_kernel" in name:
                calls[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {}
    for key, v in calls.items():
        out[key] = [statistics.mean(v[i:i + per_matrix][3:]) for i in range(0, len(v), per_matrix)]     # (warm-up calls dropped)
    return out


def main():
    dirs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "kernel_ab_*_*")))
    for d in dirs:
        trace = os.path.join(d, "r_kernel_trace.csv")
        if not os.path.isdir(d) or not os.path.exists(trace):
            continue
        parts = split(trace)
        text = "  ".join("%s %s" % (k.replace("tiled_", ""), "/".join("%.1f" % x for x in v)) for k, v in sorted(parts.items()))
        sums = [sum(x) for x in zip(*[v for k, v in sorted(parts.items()) if len(v) == len(next(iter(parts.values())))])]
        print("%-28s %s   sum %s" % (os.path.basename(d), text, "/".join("%.1f" % x for x in sums)))


if __name__ == "__main__":
    main()
