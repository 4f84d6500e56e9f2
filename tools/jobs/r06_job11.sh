#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu --durations=40 > gpurun_out/r06_j11_suite.log 2>&1
rc=$?; tail -52 gpurun_out/r06_j11_suite.log; exit $rc
