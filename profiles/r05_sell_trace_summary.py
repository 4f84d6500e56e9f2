#!/usr/bin/env python3
"""Cut a rocprofv3 kernel trace of profiles/r05_sell_in_solve.py into its phases and report k_sell's duration per phase, and
inside the solve per preceding kernel.    python profiles/r05_sell_trace_summary.py TRACE_DIR PHASES.json LABEL"""
import csv
import glob
import json
import statistics
import sys


def main(trace_dir, phases_json, label):
    meta = json.load(open(phases_json))
    files = glob.glob(f"{trace_dir}/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    short = lambda k: k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]      # noqa: E731
    sell = [(i, (e - s) / 1e3) for i, (s, e, k) in enumerate(rows) if "k_sell" in k]
    bytes_ = meta["algorithmic_bytes"]
    print(f"# {label}: {len(rows)} dispatches, {len(sell)} k_sell launches; shell CSR n = {meta['n']}, nnz = {meta['nnz']}, "
          f"algorithmic {bytes_ / 1e6:.1f} MB per launch")
    pos = 0
    for name, count in meta["phases"]:
        chunk = sell[pos:] if name == "in_solve" else sell[pos: pos + count]     # (the solve also launches look-ahead products)
        pos += len(chunk)
        if name == "warmup" or not chunk:
            continue
        d = [t for _, t in chunk]
        med = statistics.median(d)
        print(f"{label:8s} {name:14s} n={len(d):3d}  median {med:7.2f} us  mean {statistics.mean(d):7.2f}  min {min(d):7.2f}  max {max(d):7.2f}"
              f"   = {bytes_ / (med * 1e-6) / 8e12:.3f} of 8 TB/s")
        if name == "in_solve":
            by_prev = {}
            for i, t in chunk:
                prev = short(rows[i - 1][2]) if i else "-"
                gap = (rows[i][0] - rows[i - 1][1]) / 1e3 if i else 0.0
                by_prev.setdefault(prev, []).append((t, gap))
            for prev, v in sorted(by_prev.items(), key=lambda kv: -len(kv[1])):
                ts = [a for a, _ in v]
                print(f"{label:8s}   preceded by {prev[:60]:60s} n={len(v):3d}  median {statistics.median(ts):7.2f} us  "
                      f"(idle gap before it: median {statistics.median([g for _, g in v]):6.2f} us)")
    assert pos == len(sell), (pos, len(sell))


if __name__ == "__main__":
    main(*sys.argv[1:4])
