#!/usr/bin/env python3
"""Print VGPR / scratch / occupancy per kernel from hipcc's -Rpass-analysis=kernel-resource-usage.

    python arnoldi-py_amd/csrc/resource_table.py
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "aks_kernels.hip")
INC = os.path.join(HERE, "..", "..", "include")
CXXFILT = "c++filt"


def main():
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950",
           "-I" + INC, "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", SRC]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    keys = [("VGPRs", "VGPR"), ("AGPRs", "AGPR"), ("SGPRs", "SGPR"),
            (r"ScratchSize \[bytes/lane\]", "scratch"), (r"Occupancy \[waves/SIMD\]", "occ"),
            (r"LDS Size \[bytes/block\]", "LDS")]
    for blk in re.split(r"remark: Function Name: ", txt)[1:]:
        name = blk.split()[0]
        dem = subprocess.run([CXXFILT, name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        row = [f"{dem:42s}"]
        for pat, label in keys:
            m = re.search(pat + r": (\d+)", blk)
            row.append(f"{label} {m.group(1) if m else '?':>5}")
        print("  ".join(row))


if __name__ == "__main__":
    sys.exit(main())
