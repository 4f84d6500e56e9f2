#!/usr/bin/env python3
"""Summarise the rocprofv3 passes written by collect_pmc.sh into one JSON document.

Per kernel family (template arguments stripped): launches, average duration (kernel trace),
and per-launch averages of every collected counter.  HBM traffic per launch follows
MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB-units of 1024 B as printed by
rocprofv3 (x1024 -> bytes); on gfx950 FETCH_SIZE counts 128-B fabric reads as 64 B, so for a
wide coalesced stream the read bytes are 2 x FETCH_SIZE x 1024.  Both the raw and the
corrected figure are reported; which one applies to a kernel depends on its access width
(16-B gathers issue 64-B requests, which FETCH_SIZE counts exactly).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def family(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    m = re.match(r"(k_[a-z0-9_]+)(<.*>)?", name)
    if not m:
        return name[:40]
    fam = m.group(1)
    if fam == "k_spmv":
        return "k_spmv"
    return fam


def source_stamp():
    """Same stamp as bench.py: sha256 over the kernel source and the ABI header of the build the counters
    were collected on (bench.py refuses a summary whose stamp differs from the tree it runs in)."""
    import hashlib

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for rel in ("arnoldi-py_amd/csrc/aks_kernels.hip", "include/arnoldi_hip.h"):
        with open(os.path.join(repo, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def main(root):
    out = defaultdict(lambda: {"launches": 0})
    # durations
    for path in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
        dur = defaultdict(list)
        for r in csv.DictReader(open(path)):
            dur[family(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in dur.items():
            out[k]["launches"] = len(v)
            out[k]["avg_us"] = round(sum(v) / len(v) / 1e3, 2)
            out[k]["total_ms"] = round(sum(v) / 1e6, 3)
    # counters
    for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(path)):
            acc[family(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, ctrs in acc.items():
            for c, vals in ctrs.items():
                out[k][c + "_per_launch"] = round(sum(vals) / len(vals), 1)
    for k, d in out.items():
        f, w = d.get("FETCH_SIZE_per_launch"), d.get("WRITE_SIZE_per_launch")
        if f is not None and w is not None:
            d["hbm_bytes_per_launch_raw"] = int((f + w) * 1024)
            d["hbm_bytes_per_launch_fetch_x2"] = int((2 * f + w) * 1024)
        busy, act = d.get("SQ_VALU_MFMA_BUSY_CYCLES_per_launch"), d.get("GRBM_GUI_ACTIVE_per_launch")
        if busy is not None and act:
            # MfmaUtil as rocprofv3 defines it: busy cycles summed over SIMDs / (active cycles x SIMDs);
            # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS note)
            d["mfma_util_pct"] = round(100.0 * busy / ((act / 8.0) * 1024), 2)
        h, m = d.get("TCC_HIT_sum_per_launch"), d.get("TCC_MISS_sum_per_launch")
        if h is not None and m is not None and h + m > 0:
            d["l2_hit_rate"] = round(h / (h + m), 4)
    out = dict(out)
    out["_source_stamp"] = source_stamp()
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1])
