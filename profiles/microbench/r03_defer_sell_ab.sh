#!/bin/bash
# deferred normalisation in the sliced form (short rows): whole restarts of the Markov and Laplace workloads with
# AKS_DEFER_SCALE=0 (normalise at once) and =1, same build, interleaved
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_defer_sell_ab.txt; : > $out
for wl in "markov --rows 10000000 --steps 10 --warmup 2" "laplace3d --rows 16000000 --nev 10 --max-dim 40 --steps 3 --warmup 1" "laplace2d --rows 1000000 --nev 10 --max-dim 40 --steps 10 --warmup 2"; do
  for round in 1 2; do
    for flag in 0 1; do
      AKS_DEFER_SCALE=$flag timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-real-leg --no-workloads \
          > gpurun_out/ds_$flag.json 2> gpurun_out/ds_$flag.err || { echo "$flag FAILED" >> $out; tail -3 gpurun_out/ds_$flag.err >> $out; exit 1; }
      python3 - $flag "$wl" $round >> $out <<'PY'
import json, sys
flag, wl, rnd = sys.argv[1:4]
d = json.loads(open(f"gpurun_out/ds_{flag}.json").read().strip().splitlines()[-1])
o = d.get("roofline_ortho", {})
print(f"{wl.split()[0]:10s} round {rnd} defer={flag} restarts/s {d['value']:8.3f}  ms/restart {d['ms_per_step']:8.3f}  ortho ms/step {o.get('avg_ms_per_step')}  spmv ms {d['roofline'].get('avg_launch_ms')}")
PY
      tail -1 $out
    done
  done
done
