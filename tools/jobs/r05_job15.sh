python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "two_host_threads or graph or stress_grid_case or lookahead or bench_contract" > gpurun_out/r05_j15a.log 2>&1
echo "rc=$?"; tail -4 gpurun_out/r05_j15a.log | cut -c1-250
for i in 1 2 3 4 5 6; do python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "two_host_threads" 2>&1 | tail -1; done
python -m pytest tests/test_gpu_reference_full.py -q -m gpu -s -k "arnoldi_expansion" 2>&1 | grep "max |H\|passed\|failed" | cut -c1-200
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "solves_without_torch" 2>&1 | tail -1
