#!/usr/bin/env python3
"""Where the time between kernels goes: reads a rocprofv3 kernel trace (csv) and prints, per kernel name, the
count, the mean duration and the mean idle gap on the device BEFORE it (end of the previous kernel -> its start).
    python profiles/trace_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys, collections
path = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:34]
    a = acc[name]
    a[0] += 1
    a[1] += (e - s) / 1e3
    if prev_end is not None and s - prev_end < 5e6:        # (skip the pauses between phases of the program)
        a[2] += max(s - prev_end, 0) / 1e3
    prev_end = max(prev_end or e, e)
tot_d = sum(a[1] for a in acc.values()); tot_g = sum(a[2] for a in acc.values())
print(f"{path}\nkernel time {tot_d/1e3:.2f} ms, idle gaps {tot_g/1e3:.2f} ms")
for name, (n, d, g) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {name:36s} x{n:5d}  {d/n:8.1f} us  gap before {g/n:7.1f} us")
