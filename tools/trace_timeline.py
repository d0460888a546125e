"""trace_timeline.py — prints the kernel timeline (start, duration, gap to the previous end) of the LAST
`count` dispatches of a rocprofv3 --kernel-trace CSV.  usage: trace_timeline.py <kernel_trace.csv> [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-count:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-60:]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  q{r.get('Queue_Id', '?')}  {name}")
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us")
