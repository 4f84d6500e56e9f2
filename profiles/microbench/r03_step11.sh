#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "deferred or binned or truncate or c_driven_path or locking" > gpurun_out/r03_s11_tests.txt 2>&1; echo "tests rc $?"; tail -15 gpurun_out/r03_s11_tests.txt
B="--steps 20 --warmup 3 --no-cpu-baseline --no-real-leg --no-workloads"
AKS_DEFER_SCALE=0 timeout -k 10 300 python bench.py $B > gpurun_out/r03_s11_nodefer.json 2> gpurun_out/r03_s11_nodefer.err; echo "nodefer rc $?"
timeout -k 10 300 python bench.py $B > gpurun_out/r03_s11_defer.json 2> gpurun_out/r03_s11_defer.err; echo "defer rc $?"
python3 - <<'PY'
import json
for n in ("nodefer", "defer"):
    d = json.loads(open(f"gpurun_out/r03_s11_{n}.json").read().strip().splitlines()[-1])
    print(n, "restarts/s", d["value"], "ms", d["ms_per_step"], "spmv ms", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"], "ortho", d["roofline_ortho"]["avg_ms_per_step"], d["roofline_ortho"]["frac"])
PY
