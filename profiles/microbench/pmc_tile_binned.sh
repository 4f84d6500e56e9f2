#!/bin/bash
# rocprofv3 counter passes over the tile-binned SpMV micro-benchmark (run through gpurun from the repo root):
#   bash profiles/microbench/pmc_tile_binned.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_tb
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BIN=$R/profiles/microbench/tile_binned_spmv
run() {
    local name=$1; shift
    timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- "$BIN" 10000000 5 prof > "$OUT/$name.log" 2>&1 \
        || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
    echo "pass $name ok"
}
run trace --kernel-trace --stats &&
run fetch --kernel-trace --pmc FETCH_SIZE &&
run write --kernel-trace --pmc WRITE_SIZE &&
run tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum &&
run ea --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for path in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"][:60]]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, c in acc.items():
    print(k, {n: round(sum(v) / len(v), 1) for n, v in c.items()}, "launches", len(c.get("dur_us", [])))
PY
