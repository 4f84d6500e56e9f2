#!/bin/bash
cd $GRAFT_REPO_ROOT
L=arnoldi-py_amd/arnoldi_amd/lib/libarnoldi_hip.so
V=profiles/microbench/variants
out=gpurun_out/r03_rowblocks_ab.txt; : > $out
for nb in 256 512 2048; do
  echo "== base (1024 row blocks) vs $nb" >> $out
  AB_WIDTHS=12,16,20 timeout -k 10 300 python profiles/ab_kernels.py $L $V/rb$nb/libarnoldi_hip.so 10000000 3 >> $out 2>&1 || exit 1
done
echo "== n = 1.25M: base vs 256 / 512" >> $out
AB_WIDTHS=12,16,20 timeout -k 10 300 python profiles/ab_kernels.py $L $V/rb256/libarnoldi_hip.so 1250000 3 >> $out 2>&1 || exit 1
AB_WIDTHS=12,16,20 timeout -k 10 300 python profiles/ab_kernels.py $L $V/rb512/libarnoldi_hip.so 1250000 3 >> $out 2>&1 || exit 1
cat $out
