#!/bin/bash
# software-pipelined k_update_proj<NC> (two register sets, loads of the next tile issued before the current one is
# worked on) against the shipped loop, one workgroup per CU
cd $GRAFT_REPO_ROOT
L=arnoldi-py_amd/arnoldi_amd/lib/libarnoldi_hip.so
V=profiles/microbench/variants
out=gpurun_out/r03_pipe_ab.txt; : > $out
for n in 10000000 1250000; do
  echo "== n = $n: shipped | pipelined (NC <= ${PIPE_MAX:-24})" >> $out
  AB_WIDTHS=${PIPE_WIDTHS:-4,6,8,10,12,13,14,16,18,20,22,24} timeout -k 10 400 python profiles/ab_kernels.py $L $V/${PIPE_VARIANT:-pipe24}/libarnoldi_hip.so $n 3 2>&1 | grep "update_project\|kernel" >> $out || exit 1
done
cat $out
