#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout -k 5 300 ./profiles/microbench/sell_spmv > gpurun_out/r03_sell_variants.txt 2>&1; echo "sell rc $?"; cat gpurun_out/r03_sell_variants.txt
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_rocsparse_prof -o t -- $R/profiles/microbench/rocsparse_crosscheck 10000000 5 0 > $R/gpurun_out/r03_rocsparse_prof.txt 2>&1; echo "rocprof rc $?")
python3 - <<'PY'
import csv, os
p = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03_rocsparse_prof/t_kernel_stats.csv")
for r in csv.DictReader(open(p)):
    if float(r["AverageNs"]) > 20000:
        print(f'{r["Name"][:120]:120s} {r["Calls"]:>5s} calls  avg {float(r["AverageNs"]) / 1e3:9.1f} us')
PY
