set -x
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "torch_free" > gpurun_out/r05_j3_tf.log 2>&1
echo rc=$?
tail -40 gpurun_out/r05_j3_tf.log | cut -c1-300
python -m pytest tests/test_gpu_reference_full.py -q -m gpu -s > gpurun_out/r05_j3_ref.log 2>&1
echo rc=$?
grep -n "restarts (reference\|NumPy on this host\|passed\|failed\|^E " gpurun_out/r05_j3_ref.log | cut -c1-300
