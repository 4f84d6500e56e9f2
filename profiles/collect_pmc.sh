#!/bin/bash
# Collect rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats   (per-kernel durations)          -> gpurun_out/prof/trace
#   2. one --pmc pass per counter group (never mixed with traces other than --kernel-trace)
# then profiles/summarize_pmc.py turns the CSVs into profiles/pmc_summary.json.
# Usage: bash profiles/collect_pmc.sh [extra bench.py args]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${AKS_PMC_OUT:-prof}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-real-leg --no-workloads --no-device-state $*"

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
# AKS_PMC_EXTRA=1: where the waves of a kernel spend their time (issue vs wait) and what the vector L1 does
if [ "${AKS_PMC_EXTRA:-0}" = "1" ]; then
    run sq --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_ANY || true
    run tcp --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum || true
    run lat --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum || true
fi
python3 "$R/profiles/summarize_pmc.py" "$OUT" > "$OUT/pmc_summary.json"
cat "$OUT/pmc_summary.json"
