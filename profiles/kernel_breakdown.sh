#!/bin/bash
# Per-kernel-family time of one bench.py run (GPU box).  Usage: bash profiles/kernel_breakdown.sh NAME [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
AKS_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kb_$NAME -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/kb_$NAME.log 2>&1
python3 - "$R/gpurun_out/kb_$NAME" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
agg = {}
for r in rows:
    n = re.sub(r"[<(].*", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:28]
    a = agg.setdefault(n, [0, 0.0]); a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"{n:30s} calls {c:6d} total_ms {t/1e6:9.2f} avg_us {t/c/1e3:8.1f}  {100*t/tot:5.1f}%")
print(f"total kernel ms {tot/1e6:.2f}")
PY
