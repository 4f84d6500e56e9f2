#!/bin/bash
# clocks, power and temperature of the GPU while the default bench's main leg runs three times back to back (read-only queries)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_clocks_under_load.txt; : > $out
( for i in $(seq 1 100); do echo "--- t=$i s $(date +%s.%N)" >> $out; timeout 5 rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -E "sclk|mclk|fclk|socclk|Power|Temperature \(Sensor (junction|memory|edge)" >> $out; sleep 1; done ) &
MON=$!
for r in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-real-leg --no-workloads > gpurun_out/cl.json 2> gpurun_out/cl.err || { echo FAILED >> $out; break; }
  python3 -c "
import json,time
d=json.loads(open('gpurun_out/cl.json').read().strip().splitlines()[-1])
print('### bench run $r done at', time.time(), d['value'], 'spmv', d['roofline']['avg_launch_ms'], 'ortho', d['roofline_ortho']['avg_ms_per_step'])" >> $out
done
kill $MON 2>/dev/null
grep -c "t=" $out; grep "###" $out; grep -A6 "t=2 s\|t=30 s\|t=60 s\|t=90 s" $out | head -60
