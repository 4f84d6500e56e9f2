#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded_full.py -x -q -m gpu -s -k "bench_" > gpurun_out/r06_j10_tests.log 2>&1
rc=$?; tail -14 gpurun_out/r06_j10_tests.log; exit $rc
