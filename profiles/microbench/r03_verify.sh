#!/bin/bash
# full GPU suite + smoke + the default bench line on the current tree
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests_e.txt 2>&1; rc=$?; tail -3 gpurun_out/r03_gpu_tests_e.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke_e.txt 2>&1 || { tail -5 gpurun_out/r03_smoke_e.txt; exit 1; }
tail -1 gpurun_out/r03_smoke_e.txt
timeout -k 10 900 python bench.py > gpurun_out/r03_e_bench.json 2> gpurun_out/r03_e_bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_e_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("frac", "avg_launch_ms", "traffic")})
print("ortho", d["roofline_ortho"]["frac"], d["roofline_ortho"]["avg_ms_per_step"])
print("real", {k: d["real_arithmetic"].get(k) for k in ("restarts_per_s", "spmv_avg_ms", "spmv_frac")})
for w in d["workloads"]:
    print(w.get("name"), {k: w.get(k) for k in ("restarts_per_s", "restarts_per_s_eager_probed", "spmv_form", "spmv_avg_ms", "spmv_frac", "ortho_frac", "error")})
print("cpu", d["cpu_baseline"].get("value"))
PY
