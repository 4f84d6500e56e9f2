#!/bin/bash
# with one workgroup per CU for the exact-width fused kernel: where does the column-split kernel still win?
cd $GRAFT_REPO_ROOT
L=arnoldi-py_amd/arnoldi_amd/lib/libarnoldi_hip.so
V=profiles/microbench/variants
out=gpurun_out/r03_exact_split_ab.txt; : > $out
for n in 10000000 1250000; do
  echo "== n=$n: shipped (split for J > 20) vs exact-width up to 32" >> $out
  AB_WIDTHS=21,22,24,26,28,30,32 timeout -k 10 400 python profiles/ab_kernels.py $L $V/exact32/libarnoldi_hip.so $n 3 >> $out 2>&1 || exit 1
  echo "== n=$n: shipped (split for 5 <= J <= 12) vs exact-width there" >> $out
  AB_WIDTHS=5,6,7,8,9,10,11,12 timeout -k 10 400 python profiles/ab_kernels.py $L $V/nosplitlow/libarnoldi_hip.so $n 3 >> $out 2>&1 || exit 1
done
grep "==\|update_project" $out
