#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "bench_preflight or multi_rank_line or c_abi" > gpurun_out/r06_j8_tests.log 2>&1
rc=$?; tail -25 gpurun_out/r06_j8_tests.log; exit $rc
