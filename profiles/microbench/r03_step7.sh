#!/bin/bash
cd $GRAFT_REPO_ROOT
V=profiles/microbench/variants
export AB_WIDTHS=20
for v in tr_pf_ns8_w1 tr_pf_ns8_w2 tr_pf_ns4; do
  echo "== base vs $v" | tee -a gpurun_out/r03_truncate_ab.txt
  timeout -k 10 400 python profiles/ab_kernels.py $V/tr_base/libarnoldi_hip.so $V/$v/libarnoldi_hip.so 10000000 3 2>&1 | grep -v "^project\|^update" | tee -a gpurun_out/r03_truncate_ab.txt
done
export AB_WIDTHS=100
echo "== n = 1M: base vs tr_pf_ns8_w1" | tee -a gpurun_out/r03_truncate_ab.txt
timeout -k 10 400 python profiles/ab_kernels.py $V/tr_base/libarnoldi_hip.so $V/tr_pf_ns8_w1/libarnoldi_hip.so 1000000 3 2>&1 | grep -v "^project\|^update" | tee -a gpurun_out/r03_truncate_ab.txt
