#!/bin/bash
# round 6, job 1: the one-shot rewrite, the engine-level capture with a ghost exchange, the pinned capture fix, the default backend
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_j1_smoke.log 2>&1 || { tail -30 gpurun_out/r06_j1_smoke.log; exit 1; }
tail -3 gpurun_out/r06_j1_smoke.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_shot or capturable or two_host_threads or graph or torch_free or solves_without_torch or c_driven_path_ranks or rccl_collectives or c_abi or lookahead" > gpurun_out/r06_j1_parity.log 2>&1
rc=$?; tail -15 gpurun_out/r06_j1_parity.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_sharded_full.py -x -q -m gpu -k "one_shot" -s > gpurun_out/r06_j1_sharded.log 2>&1
rc=$?; tail -25 gpurun_out/r06_j1_sharded.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --rows 2000000 --steps 5 --warmup 2 --no-workloads --no-cpu-baseline --no-real-leg > gpurun_out/r06_j1_bench_small.json 2> gpurun_out/r06_j1_bench_small.err
rc=$?; tail -c 3000 gpurun_out/r06_j1_bench_small.json; tail -5 gpurun_out/r06_j1_bench_small.err; exit $rc
