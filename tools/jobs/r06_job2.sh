#!/bin/bash
# round 6, job 2: the whole GPU suite as the driver runs it (first time with the default-backend second pass and the new bench legs)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1150 python -m pytest tests/ -x -q -m gpu --durations=25 > gpurun_out/r06_j2_suite.log 2>&1
rc=$?; tail -45 gpurun_out/r06_j2_suite.log; exit $rc
