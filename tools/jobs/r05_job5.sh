set -x
python -m pytest tests/test_gpu_sharded_full.py -q -m gpu -x -s -k "one_shot or bench_eight" > gpurun_out/r05_j5_oneshot.log 2>&1
echo rc=$?
tail -30 gpurun_out/r05_j5_oneshot.log | cut -c1-400
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "one_shot" > gpurun_out/r05_j5_proc.log 2>&1
echo rc=$?
tail -30 gpurun_out/r05_j5_proc.log | cut -c1-400
