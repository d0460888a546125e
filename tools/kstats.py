"""kstats.py — short table from a rocprofv3 --stats kernel_stats.csv: kernel, calls, average us, total ms."""
import csv
import re
import sys

for path in sys.argv[1:]:
    print("#", path)
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::|spmv::detail::|void ", "", r["Name"]).split("(")[0]
        print(f"{name[:70]:70s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:10.1f} us  total {float(r['TotalDurationNs']) / 1e6:9.2f} ms")
