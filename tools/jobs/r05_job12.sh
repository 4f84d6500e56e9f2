for v in forkjoin_kernel p2p_same_stream allreduce_forked p2p_forked; do
  timeout -k 5 120 python3 profiles/r05_capture_crash_variants.py $v > gpurun_out/capvar_$v.log 2>&1
  echo "$v: exit $?"; grep "\[variant\|Fatal\|Error\|error" gpurun_out/capvar_$v.log | tail -4 | cut -c1-250
done > gpurun_out/r05_capture_variants.txt 2>&1
cat gpurun_out/r05_capture_variants.txt
python -c "import torch; print('torch', torch.__version__, 'hip', torch.version.hip)"
ls /usr/local/lib/python3.10/dist-packages/torch/lib | grep -i "amdhip\|rccl"
echo "=== c2full against the reference + full gpu suite"
python -m pytest tests -q -m gpu -x > gpurun_out/r05_j12_full.log 2>&1
echo "full suite rc=$?"; tail -6 gpurun_out/r05_j12_full.log | cut -c1-300
