set -x
python -m pytest tests/test_gpu_reference_full.py tests/test_gpu_sharded_full.py tests/test_harness.py -q -m gpu -s > gpurun_out/r05_j2_tests.log 2>&1
echo rc=$?
grep -n "restarts (reference\|NumPy on this host\|passed\|failed\|^E " gpurun_out/r05_j2_tests.log | cut -c1-400
