#!/bin/bash
cd $GRAFT_REPO_ROOT
L=arnoldi-py_amd/arnoldi_amd/lib/libarnoldi_hip.so
V=profiles/microbench/variants
out=gpurun_out/r03_rowblocks_ab2.txt; : > $out
for n in 10000000 1250000; do
  for v in rb256 rb384; do
    echo "== n=$n base (1024 row blocks) vs $v" >> $out
    AB_WIDTHS=2,4,6,8,10,11,13,14,18,21,24,32,40 timeout -k 10 400 python profiles/ab_kernels.py $L $V/$v/libarnoldi_hip.so $n 3 >> $out 2>&1 || exit 1
  done
done
grep -v truncate $out
