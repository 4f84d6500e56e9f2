python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded_full.py -q -m gpu -x -k "one_shot" > gpurun_out/r05_j14.log 2>&1
echo "rc=$?"; tail -6 gpurun_out/r05_j14.log | cut -c1-300
