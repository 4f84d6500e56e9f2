for v in forkjoin_kernel p2p_same_stream allreduce_forked p2p_forked; do
  timeout -k 5 120 python3 profiles/r05_capture_crash_variants.py $v > gpurun_out/capvar_$v.log 2>&1
  echo "$v: exit $?"; grep "\[variant\|Fatal Python\|Segmentation\|Error" gpurun_out/capvar_$v.log | tail -4 | cut -c1-250
done > gpurun_out/r05_capture_variants.txt 2>&1
cat gpurun_out/r05_capture_variants.txt
