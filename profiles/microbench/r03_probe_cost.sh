#!/bin/bash
# what the probe's event pairs cost the timed region: the default bench, every restart probed / every 4th / every 20th
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_probe_cost.txt; : > $out
for round in 1 2 3; do
  for k in 1 4 20; do
    timeout -k 10 300 python bench.py --steps 20 --warmup 3 --probe-every $k --no-cpu-baseline --no-real-leg --no-workloads > gpurun_out/pc.json 2> gpurun_out/pc.err || { echo FAILED >> $out; exit 1; }
    python3 - $k $round >> $out <<'PY'
import json, sys
d = json.loads(open("gpurun_out/pc.json").read().strip().splitlines()[-1])
print(f"round {sys.argv[2]} probe every {sys.argv[1]:>2s} restart(s): {d['value']:7.3f} restarts/s  {d['ms_per_step']:7.3f} ms  spmv {d['roofline']['avg_launch_ms']} ms over {d['roofline']['launches']} launches  ortho {d['roofline_ortho']['avg_ms_per_step']} ms/step")
PY
    tail -1 $out
  done
done
