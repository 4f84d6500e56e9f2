#!/bin/bash
# A/B of the launch geometry of the exact-width panel kernels (one workgroup per CU against 1024 workgroups): whole
# restarts through bench.py, builds interleaved, three rounds, n = 10M and the 8-GPU shard size
cd $GRAFT_REPO_ROOT
V=profiles/microbench/variants
out=gpurun_out/r03_panel_grid_ab.txt; : > $out
for n in 10000000 1250000; do
  for round in 1 2 3; do
    for lib in grid1024 percu; do
      if [ $lib = percu ]; then path=arnoldi-py_amd/arnoldi_amd/lib/libarnoldi_hip.so; else path=$V/$lib/libarnoldi_hip.so; fi
      AKS_LIB_PATH=$PWD/$path timeout -k 10 300 python bench.py --rows $n --steps 20 --warmup 3 --no-cpu-baseline --no-real-leg --no-workloads \
          > gpurun_out/pg_$lib.json 2> gpurun_out/pg_$lib.err || { echo "$lib n=$n FAILED" >> $out; tail -3 gpurun_out/pg_$lib.err >> $out; exit 1; }
      python3 - $lib $n $round >> $out <<'PY'
import json, sys
lib, n, rnd = sys.argv[1:4]
d = json.loads(open(f"gpurun_out/pg_{lib}.json").read().strip().splitlines()[-1])
o = d.get("roofline_ortho", {})
print(f"n={n:>9s} round {rnd} {lib:9s} restarts/s {d['value']:8.2f}  ms/restart {d['ms_per_step']:7.3f}  ortho ms/step {o.get('avg_ms_per_step')}  spmv ms {d['roofline'].get('avg_launch_ms')}")
PY
      tail -1 $out
    done
  done
done
