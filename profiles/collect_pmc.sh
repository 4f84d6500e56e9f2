#!/bin/bash
# Collect rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats   (per-kernel durations)          -> gpurun_out/prof/trace
#   2. one --pmc pass per counter group (never mixed with traces other than --kernel-trace)
# then profiles/summarize_pmc.py turns the CSVs into profiles/pmc_summary.json.
# Usage: bash profiles/collect_pmc.sh [extra bench.py args]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-real-leg --no-workloads $*"

run() {  # name, rocprof args...
    local name=$1; shift
    timeout -k 10 400 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 "$R/bench.py" $ARGS \
        > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
    echo "pass $name ok"
}

run trace --kernel-trace --stats &&
run fetch --kernel-trace --pmc FETCH_SIZE &&
run write --kernel-trace --pmc WRITE_SIZE &&
run tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum &&
run ea --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum &&
run req --kernel-trace --pmc TCC_REQ_sum TCC_READ_sum &&
run mfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE
python3 "$R/profiles/summarize_pmc.py" "$OUT" > "$OUT/pmc_summary.json"
cat "$OUT/pmc_summary.json"
