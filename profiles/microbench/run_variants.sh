#!/bin/bash
# Per-kernel times of pb_abi_bench under rocprofv3 for library builds given as directories holding a
# libarnoldi_hip.so (default: the in-tree build).  Usage (repo root, through gpurun):
#   ./profiles/microbench/run_variants.sh [dir ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ $# -eq 0 ] && set -- arnoldi-py_amd/arnoldi_amd/lib
for v in "$@"; do
  name=$(echo "$v" | tr '/' '_')
  export LD_LIBRARY_PATH=$R/$v:/opt/rocm/lib
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/var_$name -o t -- $R/profiles/microbench/pb_abi_bench > $R/gpurun_out/var_$name.txt 2>&1 || echo "(bench rc $?)"
  echo "== $v"; grep "tile-binned\|levels\|max |" $R/gpurun_out/var_$name.txt
  python3 - "$R/gpurun_out/var_$name/t_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_pb_phase" in r["Name"]:
        print("  ", r["Name"][28:62], r["Calls"], "calls, avg", round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
exit 0
