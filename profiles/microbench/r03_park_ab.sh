#!/bin/bash
cd $GRAFT_REPO_ROOT
L=arnoldi-py_amd/arnoldi_amd/lib/libarnoldi_hip.so
V=profiles/microbench/variants
out=gpurun_out/r03_park_ab.txt; : > $out
for v in $PARK_VARIANTS; do
  echo "== base vs $v" >> $out
  AB_WIDTHS=12,16,20 timeout -k 10 300 python profiles/ab_kernels.py $L $V/$v/libarnoldi_hip.so 10000000 3 >> $out 2>&1 || exit 1
done
grep -v truncate $out
