#!/bin/bash
# the exact-width fused kernel up to 40 columns against the previous switch points (split for 5..12 and > 20): per launch
# for J = 33..40, then whole restarts of the m = 40 / 41 workloads and of config 5
cd $GRAFT_REPO_ROOT
L=arnoldi-py_amd/arnoldi_amd/lib/libarnoldi_hip.so
V=profiles/microbench/variants
out=gpurun_out/r03_exact_split_ab2.txt; : > $out
echo "== per launch, n = 10M: previous switch points | exact-width everywhere" >> $out
AB_WIDTHS=33,34,36,38,40 timeout -k 10 400 python profiles/ab_kernels.py $V/splitold/libarnoldi_hip.so $L 10000000 3 2>&1 | grep "update_project\|kernel" >> $out || exit 1
for wl in "laplace3d --rows 16000000 --nev 10 --max-dim 40 --steps 3 --warmup 1" "laplace2d --rows 1000000 --nev 10 --max-dim 40 --steps 10 --warmup 2" "banded --rows 1500000 --per-row 35 --nev 20 --max-dim 41 --steps 10 --warmup 2" "random --steps 20 --warmup 3"; do
  for round in 1 2; do
    for lib in splitold new; do
      if [ $lib = new ]; then path=$L; else path=$V/$lib/libarnoldi_hip.so; fi
      AKS_LIB_PATH=$PWD/$path timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-real-leg --no-workloads \
          > gpurun_out/es_$lib.json 2> gpurun_out/es_$lib.err || { echo "$lib FAILED" >> $out; tail -3 gpurun_out/es_$lib.err >> $out; exit 1; }
      python3 - $lib "$wl" $round >> $out <<'PY'
import json, sys
lib, wl, rnd = sys.argv[1:4]
d = json.loads(open(f"gpurun_out/es_{lib}.json").read().strip().splitlines()[-1])
o = d.get("roofline_ortho", {})
print(f"{wl.split()[0]:10s} round {rnd} {lib:8s} restarts/s {d['value']:8.3f}  ms/restart {d['ms_per_step']:8.3f}  ortho ms/step {o.get('avg_ms_per_step')}  spmv ms {d['roofline'].get('avg_launch_ms')}")
PY
      tail -1 $out
    done
  done
done
