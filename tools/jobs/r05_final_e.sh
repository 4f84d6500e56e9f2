python bench.py > gpurun_out/r05_f_bench.json 2> gpurun_out/r05_f_bench.err
echo "bench rc=$?"; python - <<'PY'
import json
p=json.loads(open("gpurun_out/r05_f_bench.json").read().strip().splitlines()[-1])
print({k:p[k] for k in ("value","ms_per_step")}, p["roofline"]["frac"], p["roofline"]["traffic"], p["roofline"]["traffic_source"][:40], p["roofline_ortho"]["frac"])
PY
python -m pytest tests -q -m gpu -x > gpurun_out/r05_final_gpu_suite.log 2>&1
echo "suite rc=$?"; tail -6 gpurun_out/r05_final_gpu_suite.log | cut -c1-300
