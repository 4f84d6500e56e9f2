#!/bin/bash
# rocSPARSE cross-check, one algorithm per process, small size first; stops at the first failure
cd $GRAFT_REPO_ROOT
X=./profiles/microbench/rocsparse_crosscheck
O=gpurun_out/r03_rocsparse_crosscheck.txt
: > $O
timeout -k 5 120 $X 1000000 5 9 >> $O 2>&1 && \
timeout -k 5 120 $X 1000000 5 0 >> $O 2>&1 && \
timeout -k 5 120 $X 10000000 5 0 >> $O 2>&1 && \
timeout -k 5 120 $X 10000000 5 1 >> $O 2>&1 && \
timeout -k 5 120 $X 10000000 5 2 >> $O 2>&1 && \
timeout -k 5 120 $X 10000000 5 3 >> $O 2>&1 && \
timeout -k 5 120 $X 10000000 5 4 >> $O 2>&1
echo "chain rc $?" >> $O
cat $O
