"""pmc_quick.py — per-kernel means of the counters collected by tools/pmc_quick.sh."""
import collections
import csv
import glob
import os
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in glob.glob(os.path.join(sys.argv[1], "*", "*_counter_collection.csv")):
    for row in csv.DictReader(open(path)):
        k = re.sub(r"\(anonymous namespace\)::|spmv::detail::|void ", "", row["Kernel_Name"]).split("(")[0]
        if "tiled_" in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for path in glob.glob(os.path.join(sys.argv[1], "*", "*_kernel_trace.csv")):
    for row in csv.DictReader(open(path)):
        k = re.sub(r"\(anonymous namespace\)::|spmv::detail::|void ", "", row["Kernel_Name"]).split("(")[0]
        if "tiled_" in k:
            dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    print(k, "avg %.1f us (under the profiler)" % (sum(dur[k]) / max(len(dur[k]), 1)))
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-36s %16.1f" % (c, sum(v) / len(v)))
    a = acc[k]
    if "TCC_EA0_RDREQ_sum" in a and "TCC_EA0_RDREQ_LEVEL_sum" in a:
        req = sum(a["TCC_EA0_RDREQ_sum"]) / len(a["TCC_EA0_RDREQ_sum"])
        lvl = sum(a["TCC_EA0_RDREQ_LEVEL_sum"]) / len(a["TCC_EA0_RDREQ_LEVEL_sum"])
        print("   -> mean EA read latency %.0f TCC cycles, %.2f M read requests" % (lvl / max(req, 1), req / 1e6))
