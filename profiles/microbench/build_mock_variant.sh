#!/bin/bash
# libarnoldi_hip.so against the shared-memory RCCL stand-in (tests/mock_rccl) with extra -D flags, into
# profiles/microbench/variants/<name>/ (for thread-rank / shared-GPU experiments with A/B kernels).
#   ./profiles/microbench/build_mock_variant.sh <name> [-DAKS_...=...]
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
mkdir -p $R/profiles/microbench/variants/$name
make -s -C $R/tests/mock_rccl $R/tests/mock_rccl/libaksmockrccl.so
cp $R/tests/mock_rccl/libaksmockrccl.so $R/profiles/microbench/variants/$name/
NAMES="ncclGetErrorString ncclGetUniqueId ncclCommInitRank ncclCommDestroy ncclGroupStart ncclGroupEnd ncclSend ncclRecv ncclAllReduce"
REN=""; for n in $NAMES; do REN="$REN -D$n=mock_$n"; done
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -shared --offload-arch=gfx950 -I$R/include $REN -Wno-unused-function "$@" \
    -o $R/profiles/microbench/variants/$name/libarnoldi_hip.so $R/arnoldi-py_amd/csrc/aks_kernels.hip \
    -L$R/profiles/microbench/variants/$name -laksmockrccl -Wl,-rpath,'$ORIGIN'
echo "built mock variant $name: $*"
