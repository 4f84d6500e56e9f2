python bench.py > gpurun_out/r05_f_bench.json 2> gpurun_out/r05_f_bench.err
echo "bench rc=$?"; python - <<'PY'
import json
p=json.loads(open("gpurun_out/r05_f_bench.json").read().strip().splitlines()[-1])
print({k:p[k] for k in ("value","ms_per_step")}, p["roofline"]["frac"], p["roofline"]["traffic"], p["roofline"]["traffic_source"][:60], p["roofline_ortho"]["frac"], p["roofline_ortho"].get("traffic_over_algorithmic"))
for w in p["workloads"]: print(w["name"], w.get("restarts_per_s"), w.get("spmv_frac"), w.get("ortho_frac"), w.get("spmv_traffic_bytes"))
PY
python -m pytest tests/test_gpu_reference_full.py tests/test_gpu_parity.py -q -m gpu -x -k "stress_grid_matches or preflight_probes" > gpurun_out/r05_fc_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r05_fc_tests.log | cut -c1-300
