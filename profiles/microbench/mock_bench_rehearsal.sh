#!/bin/bash
# rehearsal of the driver's multi-GPU bench line on ONE GPU: N ranks share it, the library's collectives go through the
# shared-memory RCCL stand-in (tests/mock_rccl), torch.distributed (gloo) only carries the set-up exchanges
cd $GRAFT_REPO_ROOT
make -C tests/mock_rccl > /dev/null 2>&1
export AKS_LIB_PATH=$GRAFT_REPO_ROOT/tests/mock_rccl/libarnoldi_hip.so AKS_COMM_OVER_GLOO=1 AKS_BENCH_BACKEND=gloo AKS_GRAPH=0
for N in 2 4; do
  timeout -k 10 400 python bench.py --gpus $N --rows 1000000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_s12_mock_$N.json 2> gpurun_out/r03_s12_mock_$N.err; echo "N=$N rc $?"
  python3 - $N <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r03_s12_mock_{n}.json").read().strip().splitlines()[-1])
    print("n_gpus", d["n_gpus"], "value", d["value"], "path:", d["config"]["path"])
    print("  exchange:", d["config"]["exchange"])
    print("  roofline:", {k: d["roofline"][k] for k in ("kernel", "avg_launch_ms", "frac")})
except Exception as e:
    print("failed", e); print(open(f"gpurun_out/r03_s12_mock_{n}.err").read()[-2500:])
PY
done
