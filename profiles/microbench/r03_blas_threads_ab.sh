#!/bin/bash
# A/B of the host BLAS thread limit (arnoldi_amd.utils.host_blas_threads): whole restarts through bench.py, same build,
# processes interleaved, AKS_HOST_BLAS_THREADS=keep (the pools as numpy/scipy set them up) against the default (1).
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_blas_threads_ab.txt; : > $out
for n in 1250000 2000000 10000000; do
  for round in 1 2 3; do
    for mode in keep 1; do
      AKS_HOST_BLAS_THREADS=$mode timeout -k 10 300 python bench.py --rows $n --steps 20 --warmup 3 --no-cpu-baseline --no-real-leg --no-workloads \
          > gpurun_out/bt_$mode.json 2> gpurun_out/bt_$mode.err || { echo "$mode n=$n FAILED" >> $out; exit 1; }
      python3 - $mode $n $round >> $out <<'PY'
import json, sys
mode, n, rnd = sys.argv[1:4]
d = json.loads(open(f"gpurun_out/bt_{mode}.json").read().strip().splitlines()[-1])
o = d.get("roofline_ortho", {})
print(f"n={n:>9s} round {rnd} host BLAS threads {mode:5s} restarts/s {d['value']:8.2f}  ms/restart {d['ms_per_step']:7.3f}  ortho ms/step {o.get('avg_ms_per_step')}  spmv ms {d['roofline'].get('avg_launch_ms')}  graph replay {d['config'].get('restarts_per_s_hipgraph')}")
PY
      tail -1 $out
    done
  done
done
