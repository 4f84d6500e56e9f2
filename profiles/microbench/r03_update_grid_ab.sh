#!/bin/bash
# A/B of the launch geometry of the second-pass update kernel k_update<true> (every step of the Laplacians): whole
# restarts of the 2-D and 3-D Laplace workloads, builds interleaved
cd $GRAFT_REPO_ROOT
V=profiles/microbench/variants
out=gpurun_out/r03_update_grid_ab.txt; : > $out
for wl in "laplace3d --rows 16000000 --nev 10 --max-dim 40 --steps 3 --warmup 1" "laplace2d --rows 1000000 --nev 10 --max-dim 40 --steps 10 --warmup 2"; do
  for round in 1 2; do
    for lib in base upd256 upd512; do
      if [ $lib = base ]; then path=arnoldi-py_amd/arnoldi_amd/lib/libarnoldi_hip.so; else path=$V/$lib/libarnoldi_hip.so; fi
      AKS_LIB_PATH=$PWD/$path timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-real-leg --no-workloads \
          > gpurun_out/ug_$lib.json 2> gpurun_out/ug_$lib.err || { echo "$lib FAILED" >> $out; tail -3 gpurun_out/ug_$lib.err >> $out; exit 1; }
      python3 - $lib "$wl" $round >> $out <<'PY'
import json, sys
lib, wl, rnd = sys.argv[1:4]
d = json.loads(open(f"gpurun_out/ug_{lib}.json").read().strip().splitlines()[-1])
o = d.get("roofline_ortho", {})
print(f"{wl.split()[0]:10s} round {rnd} {lib:7s} restarts/s {d['value']:8.3f}  ms/restart {d['ms_per_step']:8.3f}  ortho ms/step {o.get('avg_ms_per_step')}  spmv ms {d['roofline'].get('avg_launch_ms')}")
PY
      tail -1 $out
    done
  done
done
