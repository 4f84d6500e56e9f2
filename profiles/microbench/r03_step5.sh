#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rccl_collectives_one_rank or c_driven_path or two_ranks_share" > gpurun_out/r03_s5_tests.txt 2>&1; echo "tests rc $?"; tail -25 gpurun_out/r03_s5_tests.txt
