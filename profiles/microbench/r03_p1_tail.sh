#!/bin/bash
# does the last, partly filled round of sub-slab workgroups cost phase 1?  1221 sub-slabs (n = 10M: 4.77 rounds of 256)
# against 1280 (n = 10485760: 5.00 rounds) and 1024 (n = 8388608: 4.00 rounds); kernel times from rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03_p1_tail.txt; : > $out
for n in 10000000 10485760 8388608; do
  rm -rf $R/gpurun_out/p1tail
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p1tail -- python3 $R/bench.py --rows $n --steps 5 --warmup 1 --no-cpu-baseline --no-real-leg --no-workloads > $R/gpurun_out/p1tail.log 2>&1 || { tail -3 $R/gpurun_out/p1tail.log; exit 1; }
  python3 - $n $R >> $out <<'PY'
import csv, glob, sys
n, R = int(sys.argv[1]), sys.argv[2]
f = glob.glob(R + "/gpurun_out/p1tail/**/*kernel_stats.csv", recursive=True)[0]
t = {}
for r in csv.DictReader(open(f)):
    for k in ("k_pb_phase1", "k_pb_phase2"):
        if k in r["Name"]: t[k] = float(r["AverageNs"]) / 1e3
slabs = (n + 8191) // 8192
print(f"n={n:9d} sub-slabs {slabs} = {slabs/256:.2f} rounds of 256: phase 1 {t['k_pb_phase1']:7.2f} us = {t['k_pb_phase1']*1e3/(5*n):.4f} ns per non-zero, phase 2 {t['k_pb_phase2']:7.2f} us = {t['k_pb_phase2']*1e3/(5*n):.4f} ns per non-zero")
PY
  tail -1 $out
done
