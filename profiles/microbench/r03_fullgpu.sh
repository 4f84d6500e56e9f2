#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r03_gpu_tests_full.txt 2>&1; echo "tests rc $?"; tail -8 gpurun_out/r03_gpu_tests_full.txt
