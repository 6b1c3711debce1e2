#!/usr/bin/env python
"""device timeline of the LAST repetition of a kernel sequence in a rocprofv3 kernel trace: python tools/timeline.py <dir> <first-kernel-substring> [skip_last=0]"""
import csv, glob, os, re, sys
d, first = sys.argv[1], sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [k for k, r in enumerate(rows) if first in r["Kernel_Name"]]
a, b = starts[-2 - skip], starts[-1 - skip]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
busy = 0
print(f"{'kernel':70s} {'start us':>9s} {'dur us':>8s} {'gap us':>7s}")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("mdh::", "")[:68]
    print(f"{name:70s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"step = {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, {b - a} launches")
