#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_real.py -x -q -m gpu -k "locking or deflate or explicit_restarts" > gpurun_out/r03_s8_tests.txt 2>&1; echo "tests rc $?"; tail -15 gpurun_out/r03_s8_tests.txt
