#!/bin/bash
# final measurement campaign, part B: PMC passes of the other BASELINE shapes (the `workloads` legs of bench.py)
cd $GRAFT_REPO_ROOT
run() {  # name, bench args...
  local name=$1; shift
  AKS_PMC_OUT=prof_$name bash profiles/collect_pmc.sh "$@" > gpurun_out/r03_final_b_$name.log 2>&1; echo "$name pmc rc $?"
  cp gpurun_out/prof_$name/pmc_summary.json profiles/pmc_summary_$name.json
}
run markov --workload markov --rows 10000000 &&
run laplace2d --workload laplace2d --rows 1000000 --nev 10 --max-dim 40 &&
run banded --workload banded --rows 1500000 --per-row 35 --nev 20 --max-dim 41 &&
run laplace3d --workload laplace3d --rows 16000000 --nev 10 --max-dim 40
python3 - <<'PY'
import json
for w in ("markov", "laplace2d", "banded", "laplace3d"):
    d = json.load(open(f"profiles/pmc_summary_{w}.json"))
    k = d.get("k_sell", {})
    print(w, "k_sell avg_us", k.get("avg_us"), "hbm x2", k.get("hbm_bytes_per_launch_fetch_x2"), "l2 hit", k.get("l2_hit_rate"))
PY
