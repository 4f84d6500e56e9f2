#!/bin/bash
cd $GRAFT_REPO_ROOT
L=arnoldi-py_amd/arnoldi_amd/lib/libarnoldi_hip.so
V=profiles/microbench/variants
out=gpurun_out/r03_proj40_ab.txt; : > $out
for n in 10000000 1250000; do
echo "== per launch, n = $n: projection in groups of <= 32 columns | one launch up to 40" >> $out
AB_WIDTHS=33,34,36,38,40 timeout -k 10 400 python profiles/ab_kernels.py $L $V/proj40/libarnoldi_hip.so $n 3 2>&1 | grep "project\|kernel" | grep -v update >> $out || exit 1
done
cat $out
