#!/bin/bash
cd $GRAFT_REPO_ROOT
L=arnoldi-py_amd/arnoldi_amd/lib/libarnoldi_hip.so
V=profiles/microbench/variants
out=gpurun_out/r03_trunc_grid_ab.txt; : > $out
for n in 10000000 1250000; do
for k in 256 512 1024 2048; do
  echo "== n=$n base (4096 workgroups) vs $k" >> $out
  AB_WIDTHS=41 timeout -k 10 300 python profiles/ab_kernels.py $L $V/trunc$k/libarnoldi_hip.so $n 3 >> $out 2>&1 || exit 1
done
done
grep "==\|truncate" $out
