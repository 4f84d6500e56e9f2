#!/bin/bash
# round-3 step 1: correctness of the queue-based phase 2 on the device, then per-kernel times of the build variants
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spmv or binned" > gpurun_out/r03_s1_tests.txt 2>&1; echo "tests rc $?" ; tail -3 gpurun_out/r03_s1_tests.txt
./profiles/microbench/run_variants.sh arnoldi-py_amd/arnoldi_amd/lib profiles/microbench/variants/k8q128 2>&1 | tee gpurun_out/r03_s1_variants.txt
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/profiles/microbench/variants/ticks:/opt/rocm/lib
PB_TICKS=1 timeout -k 10 120 ./profiles/microbench/pb_abi_bench 2>&1 | tee gpurun_out/r03_s1_ticks.txt
