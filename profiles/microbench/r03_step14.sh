#!/bin/bash
# the preflight under torch.distributed.run (as the driver launches N > 1), one rank, forced communicator
cd $GRAFT_REPO_ROOT
AKS_FORCE_COMM=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 1 --rows 1250000 --steps 5 --warmup 2 --no-cpu-baseline --no-real-leg --no-workloads > gpurun_out/r03_s14.json 2> gpurun_out/r03_s14.err; echo "rc $?"
python3 - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r03_s14.json").read().splitlines() if l.startswith("{")][-1])
    print("value", d["value"], "path:", d["config"]["path"]); print("preflight:", d["config"]["native_preflight"])
except Exception as e:
    print("failed", e); print(open("gpurun_out/r03_s14.err").read()[-3000:])
PY
