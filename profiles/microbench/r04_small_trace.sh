#!/bin/bash
# kernel timeline of the restarts at the 8-GPU shard size (n = 1.25M): per-kernel durations and the idle gaps between
# consecutive kernels on the device, eager and as a hipGraph
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_small_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/bench.py --rows 1250000 --steps 10 --warmup 2 --no-cpu-baseline --no-real-leg --no-workloads > $OUT/t.log 2>&1 || { tail -5 $OUT/t.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def name(r):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    if k.startswith("void "):
        k = k[5:]
    return k.split("(")[0][:40]
# steady state: take the last 60 % of the trace
rows = rows[int(len(rows) * 0.4):]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    dur[name(a)].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gap[name(a) + " -> " + name(b)].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
print("kernel durations (us): count, mean")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:42s} {len(v):6d} {sum(v)/len(v)/1e3:9.2f}   total {sum(v)/1e6:8.2f} ms")
print("gaps between consecutive kernels (us): count, mean, total")
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f"  {k:70s} {len(v):6d} {sum(v)/len(v)/1e3:9.2f}   total {sum(v)/1e6:8.2f} ms")
tot = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(sum(v) for v in dur.values())
small = sum(sum(v) for k, v in dur.items() if k.startswith(("k_reduce", "k_update<", "k_finish", "k_colscale")))
print(f"span {tot/1e6:.2f} ms, kernels busy {busy/1e6:.2f} ms = {busy/tot:.3f}; the one-block / predicated-off kernels "
      f"(k_reduce, k_update<true>, k_finish, k_colscale): {small/1e6:.2f} ms = {small/busy:.3f} of the busy time")
PY
