#!/bin/bash
# round 6, job 3: the rest of the GPU suite behind the bench-preflight test (job 2 stopped there: -x)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bench_preflight or rccl_collectives or solves_without_torch or capturable or graph_replay or failed_graph or stress_grid or locking or c_abi or lookahead or bench_contract" --durations=10 > gpurun_out/r06_j3_parity.log 2>&1
rc=$?; tail -25 gpurun_out/r06_j3_parity.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 1000 python -m pytest tests/test_gpu_real.py tests/test_gpu_reference_full.py tests/test_gpu_sharded_full.py tests/test_harness.py -x -q -m gpu --durations=15 > gpurun_out/r06_j3_rest.log 2>&1
rc=$?; tail -30 gpurun_out/r06_j3_rest.log; exit $rc
