#!/bin/bash
# final measurement campaign, part A: default workload (config 5): rocprofv3 stats + PMC passes, then the full bench line
cd $GRAFT_REPO_ROOT
AKS_PMC_OUT=prof_c5 bash profiles/collect_pmc.sh > gpurun_out/r03_final_a_pmc.log 2>&1; echo "pmc rc $?"; grep "pass " gpurun_out/r03_final_a_pmc.log
cp gpurun_out/prof_c5/pmc_summary.json profiles/pmc_summary.json
timeout -k 10 900 python bench.py > gpurun_out/r03_final_bench.json 2> gpurun_out/r03_final_bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_final_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"])
print("ortho", d["roofline_ortho"]); print("restart", d["restart_roofline"])
print("real", {k: d["real_arithmetic"].get(k) for k in ("restarts_per_s", "spmv_avg_ms", "spmv_frac")})
for w in d["workloads"]:
    print(w.get("name"), {k: w.get(k) for k in ("restarts_per_s", "spmv_form", "spmv_avg_ms", "spmv_frac", "ortho_frac", "spmv_traffic_bytes", "error")})
print("cpu", d["cpu_baseline"])
PY
