#!/bin/bash
# A/B of the serpentine Gram-Schmidt sweep (AKS_GS_SERP) and of the panel load policy (AKS_NT_PANEL): whole restarts
# through bench.py on one box, builds interleaved, two rounds, three problem sizes (10M = config 5, 2M / 1.25M = the
# 8-GPU shard sizes of configs 4 and 5).
cd $GRAFT_REPO_ROOT
V=profiles/microbench/variants
out=gpurun_out/r03_serp_ab.txt; : > $out
for n in 10000000 2000000 1250000; do
  for round in 1 2; do
    for lib in base serp1 serp1_nont serp0_nont; do
      if [ $lib = base ]; then path=arnoldi-py_amd/arnoldi_amd/lib/libarnoldi_hip.so; else path=$V/$lib/libarnoldi_hip.so; fi
      AKS_LIB_PATH=$PWD/$path timeout -k 10 300 python bench.py --rows $n --steps 20 --warmup 3 --no-cpu-baseline --no-real-leg --no-workloads \
          > gpurun_out/serp_$lib.json 2> gpurun_out/serp_$lib.err || { echo "$lib n=$n FAILED" >> $out; exit 1; }
      python3 - $lib $n $round >> $out <<'PY'
import json, sys
lib, n, rnd = sys.argv[1:4]
d = json.loads(open(f"gpurun_out/serp_{lib}.json").read().strip().splitlines()[-1])
o = d.get("roofline_ortho", {})
print(f"n={n:>9s} round {rnd} {lib:11s} restarts/s {d['value']:8.2f}  ms/restart {d['ms_per_step']:7.3f}  ortho ms/step {o.get('avg_ms_per_step')}  spmv ms {d['roofline'].get('avg_launch_ms')}")
PY
      tail -1 $out
    done
  done
done
