#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_real.py tests/test_gpu_parity.py -x -q -m gpu -k "deflate or bench_multi_rank or c_driven_path" > gpurun_out/r03_s13_tests.txt 2>&1; echo "tests rc $?"; tail -12 gpurun_out/r03_s13_tests.txt
