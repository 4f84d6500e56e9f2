#!/bin/bash
# bench.py --gpus 2 at n = 10M on ONE GPU over the stand-in, with rank 0's operator set-up profiled (AKS_PROFILE_SETUP)
cd $GRAFT_REPO_ROOT
AKS_PROFILE_SETUP=1 AKS_LIB_PATH=$PWD/tests/mock_rccl/libarnoldi_hip.so AKS_COMM_OVER_GLOO=1 AKS_BENCH_BACKEND=gloo AKS_GRAPH=0 \
  timeout -k 10 600 python bench.py --gpus 2 --rows 10000000 --steps 2 --warmup 1 --no-cpu-baseline --no-workloads \
  > gpurun_out/two_rank_setup.json 2> gpurun_out/two_rank_setup.err
grep -A22 "rows built" gpurun_out/two_rank_setup.err | cut -c1-150
python - <<'PY'
import json
d = json.loads(open("gpurun_out/two_rank_setup.json").read().strip().splitlines()[-1])
print("restarts/s", d["value"], "setup_s", d.get("setup_s"))
PY
