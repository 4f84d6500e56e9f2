#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_real.py -x -q -m gpu -k "spmv or c_abi or bench_contract or c_driven_path" > gpurun_out/r03_s15_tests.txt 2>&1; echo "tests rc $?"; tail -6 gpurun_out/r03_s15_tests.txt
timeout -k 10 900 python bench.py --no-cpu-baseline > gpurun_out/r03_s15_bench.json 2> gpurun_out/r03_s15_bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_s15_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "spmv ms", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], d["roofline"]["traffic_source"])
print("ortho", d["roofline_ortho"]["avg_ms_per_step"], d["roofline_ortho"]["frac"])
for w in d["workloads"]:
    print(w.get("name"), {k: w.get(k) for k in ("restarts_per_s", "spmv_avg_ms", "spmv_frac", "spmv_traffic_bytes", "error")})
PY
