python bench.py > gpurun_out/r05_f_bench.json 2> gpurun_out/r05_f_bench.err
echo "bench rc=$?"; python - <<'PY'
import json
p=json.loads(open("gpurun_out/r05_f_bench.json").read().strip().splitlines()[-1])
print({k:p[k] for k in ("value","ms_per_step")}, p["roofline"]["frac"], p["roofline"]["traffic"], p["roofline"]["traffic_source"][:60], p["roofline_ortho"]["frac"])
for w in p["workloads"]: print(w["name"], w.get("restarts_per_s"), w.get("spmv_frac"), w.get("ortho_frac"))
print(p["cpu_baseline"].get("value"), p.get("real_arithmetic",{}).get("restarts_per_s"))
PY
bash profiles/microbench/r05_final_pmc.sh c5 markov laplace2d
