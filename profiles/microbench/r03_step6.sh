#!/bin/bash
cd $GRAFT_REPO_ROOT
B="--rows 1250000 --steps 20 --warmup 3 --no-cpu-baseline --no-real-leg --no-workloads"
timeout -k 10 300 python bench.py $B > gpurun_out/r03_s6_plain.json 2> gpurun_out/r03_s6_plain.err; echo "plain rc $?"
AKS_FORCE_COMM=1 timeout -k 10 600 python bench.py $B > gpurun_out/r03_s6_comm.json 2> gpurun_out/r03_s6_comm.err; echo "comm rc $?"
AKS_FORCE_COMM=1 AKS_GRAPH_COMM=1 AKS_BENCH_PREFLIGHT=0 timeout -k 10 300 python bench.py $B > gpurun_out/r03_s6_commgraph.json 2> gpurun_out/r03_s6_commgraph.err; echo "commgraph rc $?"
python3 - <<'PY'
import json, os
for n in ("plain", "comm", "commgraph"):
    try:
        d = json.loads(open(f"gpurun_out/r03_s6_{n}.json").read().strip().splitlines()[-1])
        print(n, "value", d["value"], "ms", d["ms_per_step"], "path:", d["config"]["path"], "| preflight:", d["config"].get("native_preflight"), "| exchange:", d["config"]["exchange"])
    except Exception as e:
        print(n, "failed:", e, open(f"gpurun_out/r03_s6_{n}.err").read()[-1500:])
PY
