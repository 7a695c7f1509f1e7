"""Summarise rocprofv3 outputs produced by tests/prof.sh: per-kernel avg duration
(kernel-trace stats) and per-kernel mean PMC counter values per dispatch."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"void (k_\w+)<(.*)>\(", name)
    if m:
        return f"{m.group(1)}<{m.group(2)[:28]}>"
    return name[:48]


for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats", f)
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print(f"{short(r['Name']):50s} calls={r['Calls']:>5s} avg_ns={float(r['AverageNs']):12.0f} "
              f"total_ms={float(r['TotalDurationNs'])/1e6:9.2f} pct={r['Percentage']}")

agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not k.startswith("k_"):
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== PMC (mean per dispatch)")
for k, cs in sorted(agg.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:24s} {sum(v)/len(v):16.1f}   (n={len(v)})")
