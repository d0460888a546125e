#!/bin/bash
# same-box A/B of engine knobs on C5 / C2: every configuration three times, interleaved; prints all and the medians
# usage: tools/ab_sweep.sh "<ENV=VAL ...>" "<ENV=VAL ...>" ...   (an empty string = defaults); WHICH="c5only c2only" by default
cd "$(dirname "$0")/.."
out=gpurun_out/ab_sweep.txt
mkdir -p gpurun_out; : > $out
WHICH=${WHICH:-"c5only c2only"}
for round in 1 2 3; do
  i=0
  for cfg in "$@"; do
    i=$((i + 1))
    env $cfg python tools/quick_bench.py $WHICH 2>&1 | grep "kernel=" | sed "s|^|cfg$i round$round [$cfg] |" >> $out
  done
done
python3 - "$out" <<'PY'
import collections, re, statistics, sys
acc = collections.defaultdict(list)
for line in open(sys.argv[1]):
    m = re.match(r"(cfg\d+) round\d+ \[(.*?)\] (.*?)\s+kernel=(\d+).*avg=\s*([\d.]+)us min=\s*([\d.]+)us", line)
    if m: acc[(m.group(1), m.group(2), m.group(3).strip() + " k" + m.group(4))].append((float(m.group(5)), float(m.group(6))))
for (cfg, env, name), v in sorted(acc.items(), key=lambda kv: (kv[0][2], int(kv[0][0][3:]))):
    avgs, mins = [a for a, _ in v], [b for _, b in v]
    print(f"{cfg} [{env or 'defaults'}] {name:22s} median avg {statistics.median(avgs):8.1f} us  best min {min(mins):8.1f} us   avgs {avgs}")
PY
