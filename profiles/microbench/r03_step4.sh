#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python profiles/sell_context_probe.py > gpurun_out/r03_sell_context.txt 2>&1; echo "probe rc $?"; tail -4 gpurun_out/r03_sell_context.txt
timeout -k 10 200 ./profiles/microbench/sell_spmv > gpurun_out/r03_sell_variants2.txt 2>&1; sed -n 14,27p gpurun_out/r03_sell_variants2.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_host_threads or c_abi or binned or truncate" > gpurun_out/r03_s4_tests.txt 2>&1; echo "tests rc $?"; tail -5 gpurun_out/r03_s4_tests.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "form_is_chosen" > gpurun_out/r03_s4_tests2.txt 2>&1; echo "tests2 rc $?"; tail -5 gpurun_out/r03_s4_tests2.txt
timeout -k 10 600 python profiles/lpr_sweep.py > gpurun_out/r03_lpr_sweep.txt 2>&1; echo "lpr rc $?"; cat gpurun_out/r03_lpr_sweep.txt
