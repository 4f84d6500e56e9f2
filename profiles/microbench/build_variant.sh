#!/bin/bash
# Builds libarnoldi_hip.so with extra -D flags into profiles/microbench/variants/<name>/ (git-ignored, shipped by gpurun).
#   ./profiles/microbench/build_variant.sh <name> [-DAKS_...=... ...]
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
mkdir -p $R/profiles/microbench/variants/$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -I$R/include "$@" \
    -o $R/profiles/microbench/variants/$name/libarnoldi_hip.so $R/arnoldi-py_amd/csrc/aks_kernels.hip -ldl -pthread
echo "built variants/$name: $*"
