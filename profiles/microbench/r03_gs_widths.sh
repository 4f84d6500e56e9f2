#!/bin/bash
# Gram-Schmidt stage kernels over the panel widths of configs 2-4 (J up to 40) at n = 10M: ms and TB/s per launch
cd $GRAFT_REPO_ROOT
L=arnoldi-py_amd/arnoldi_amd/lib/libarnoldi_hip.so
AB_WIDTHS=8,12,16,20,21,24,28,32,36,40 timeout -k 10 600 python profiles/ab_kernels.py $L $L 10000000 2 > gpurun_out/r03_gs_widths.txt 2>&1
cat gpurun_out/r03_gs_widths.txt
