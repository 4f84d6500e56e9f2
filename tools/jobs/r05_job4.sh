set -x
python -m pytest tests/test_gpu_sharded_full.py -q -m gpu -x -s -k "one_shot" > gpurun_out/r05_j4_oneshot.log 2>&1
echo rc=$?
tail -30 gpurun_out/r05_j4_oneshot.log | cut -c1-400
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "one_shot or torch_free" > gpurun_out/r05_j4_proc.log 2>&1
echo rc=$?
tail -30 gpurun_out/r05_j4_proc.log | cut -c1-400
python -m pytest tests/test_gpu_sharded_full.py -q -m gpu -x -s -k "bitwise_repeatable" > gpurun_out/r05_j4_repeat.log 2>&1
echo rc=$?
tail -12 gpurun_out/r05_j4_repeat.log | cut -c1-400
