#!/bin/bash
# SQ / LDS / TCC counters of the library's binned SpMV kernels through the C-ABI benchmark:
#   bash profiles/microbench/pmc_pb_abi.sh        (through gpurun, from the repo root)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_abi
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BIN=$R/profiles/microbench/pb_abi_bench
run() {
    local name=$1; shift
    timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- "$BIN" 10000000 5 0 0 > "$OUT/$name.log" 2>&1 \
        || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
    echo "pass $name ok"
}
run sq1 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES &&
run sq2 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM &&
run lds --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_INSTS_SALU SQ_INSTS_VALU &&
run fetch --kernel-trace --pmc FETCH_SIZE &&
run write --kernel-trace --pmc WRITE_SIZE &&
run tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"][28:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if "k_pb" in k or "k_spmv" in k:
        print(k)
        for n, v in sorted(c.items()):
            print("   %-28s %16.1f" % (n, sum(v) / len(v)))
PY
