#!/bin/bash
# the update coefficients of k_update_proj<NC>: in registers for the whole loop, or re-read from LDS per tile (the
# round-2 choice above 18 columns, made to keep two waves per SIMD) -- with one workgroup per CU
cd $GRAFT_REPO_ROOT
L=arnoldi-py_amd/arnoldi_amd/lib/libarnoldi_hip.so
V=profiles/microbench/variants
out=gpurun_out/r03_hs_ab.txt; : > $out
echo "== n = 10M: shipped (LDS above 18 columns) | registers at every width" >> $out
AB_WIDTHS=16,18,19,20,24,28,32,36,40 timeout -k 10 400 python profiles/ab_kernels.py $L $V/hsreg/libarnoldi_hip.so 10000000 3 2>&1 | grep "update_project\|kernel" >> $out || exit 1
echo "== n = 10M: shipped | LDS above 12 columns" >> $out
AB_WIDTHS=13,14,16,18 timeout -k 10 400 python profiles/ab_kernels.py $L $V/hslds12/libarnoldi_hip.so 10000000 3 2>&1 | grep "update_project\|kernel" >> $out || exit 1
cat $out
