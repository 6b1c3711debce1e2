#!/usr/bin/env python
"""print the top kernels of a rocprofv3 --stats kernel_stats.csv: python tools/kstats.py <dir-or-csv> [top] [divide_calls_by]"""
import csv, glob, os, re, sys
path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
if os.path.isdir(path):
    hits = glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True)
    if not hits:
        sys.exit(f"no kernel_stats.csv under {path}")
    path = hits[0]
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:top]:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("mdh::", "")[:64]
    print(f"{name:64s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:9.1f} us  {float(r['TotalDurationNs']) / tot * 100:5.1f} %")
