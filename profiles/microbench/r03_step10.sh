#!/bin/bash
cd $GRAFT_REPO_ROOT
B="--steps 20 --warmup 3 --no-cpu-baseline --no-real-leg --no-workloads"
for rep in 1 2; do
AKS_DEFER_SCALE=0 timeout -k 10 300 python bench.py $B > gpurun_out/r03_s10_nodefer_$rep.json 2> gpurun_out/r03_s10_nodefer.err; echo "nodefer rc $?"
timeout -k 10 300 python bench.py $B > gpurun_out/r03_s10_defer_$rep.json 2> gpurun_out/r03_s10_defer.err; echo "defer rc $?"
done
python3 - <<'PY'
import json
for n in ("nodefer_1", "defer_1", "nodefer_2", "defer_2"):
    d = json.loads(open(f"gpurun_out/r03_s10_{n}.json").read().strip().splitlines()[-1])
    print(n, "restarts/s", d["value"], "ms", d["ms_per_step"], "spmv ms", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"], "ortho", d["roofline_ortho"]["avg_ms_per_step"], d["roofline_ortho"]["frac"])
PY
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c_driven_path" > gpurun_out/r03_s10_tests.txt 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r03_s10_tests.txt
