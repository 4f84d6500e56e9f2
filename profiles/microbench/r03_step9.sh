#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "deferred or binned or c_abi or truncate or c_driven_path" > gpurun_out/r03_s9_tests.txt 2>&1; echo "tests rc $?"; tail -25 gpurun_out/r03_s9_tests.txt
