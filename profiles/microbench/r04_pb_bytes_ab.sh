#!/bin/bash
# VERDICT r03 item 5, time-boxed: sub-slabs of 10 240 columns and Infinity-Cache-resident product groups, measured
# (profiles/pb_bytes_ab.py) on one box, builds interleaved, two rounds.  Variants (profiles/microbench/build_variant.sh):
#   pbplain -DAKS_PB_NT_STORE=0 | pb10240 -DAKS_PB_SLAB_COLS=10240 | pb10240plain (both)
cd $GRAFT_REPO_ROOT
V=profiles/microbench/variants
out=gpurun_out/r04_pb_bytes_ab.txt; : > $out
for round in 1 2; do
  echo "== round $round" >> $out
  AKS_LIB_PATH=$PWD/arnoldi-py_amd/arnoldi_amd/lib/libarnoldi_hip.so timeout -k 10 300 python profiles/pb_bytes_ab.py 10000000 2,4,8 >> $out 2>&1 || exit 1
  AKS_LIB_PATH=$PWD/$V/pbplain/libarnoldi_hip.so timeout -k 10 300 python profiles/pb_bytes_ab.py 10000000 2,4,8 >> $out 2>&1 || exit 1
  AKS_LIB_PATH=$PWD/$V/pb10240/libarnoldi_hip.so timeout -k 10 300 python profiles/pb_bytes_ab.py 10000000 "" >> $out 2>&1 || exit 1
  AKS_LIB_PATH=$PWD/$V/pb10240plain/libarnoldi_hip.so timeout -k 10 300 python profiles/pb_bytes_ab.py 10000000 "" >> $out 2>&1 || exit 1
done
