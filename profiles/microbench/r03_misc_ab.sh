#!/bin/bash
# two more settings from rounds 1-2 re-checked under the new launch geometry: non-temporal panel loads (per launch), and the
# workgroup count of the normalisation pass k_finish (whole restarts of the 3-D Laplacian, which normalises at every step)
cd $GRAFT_REPO_ROOT
L=arnoldi-py_amd/arnoldi_amd/lib/libarnoldi_hip.so
V=profiles/microbench/variants
out=gpurun_out/r03_misc_ab.txt; : > $out
echo "== n = 10M per launch: shipped (non-temporal panel loads) | plain loads" >> $out
AB_WIDTHS=8,12,16,20,32,40 timeout -k 10 400 python profiles/ab_kernels.py $L $V/ntoff/libarnoldi_hip.so 10000000 3 2>&1 | grep "project\|kernel" >> $out || exit 1
for round in 1 2; do
  for lib in base fin256 fin512; do
    if [ $lib = base ]; then path=$L; else path=$V/$lib/libarnoldi_hip.so; fi
    AKS_LIB_PATH=$PWD/$path timeout -k 10 300 python bench.py --workload laplace3d --rows 16000000 --nev 10 --max-dim 40 --steps 3 --warmup 1 --no-cpu-baseline --no-real-leg --no-workloads \
        > gpurun_out/mi_$lib.json 2> gpurun_out/mi_$lib.err || { echo "$lib FAILED" >> $out; exit 1; }
    python3 - $lib $round >> $out <<'PY'
import json, sys
lib, rnd = sys.argv[1:3]
d = json.loads(open(f"gpurun_out/mi_{lib}.json").read().strip().splitlines()[-1])
o = d.get("roofline_ortho", {})
print(f"laplace3d round {rnd} k_finish workgroups {lib:7s} restarts/s {d['value']:8.3f}  ms/restart {d['ms_per_step']:8.3f}  ortho ms/step {o.get('avg_ms_per_step')}")
PY
  done
done
cat $out
