#!/bin/bash
# round-3 step 2: rocSPARSE cross-check on config 5 (plain + under rocprofv3), PMC passes for the banded workload
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout -k 10 300 ./profiles/microbench/rocsparse_crosscheck 2>&1 | tee gpurun_out/r03_rocsparse_crosscheck.txt
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_rocsparse_prof -o t -- $R/profiles/microbench/rocsparse_crosscheck > $R/gpurun_out/r03_rocsparse_prof.txt 2>&1; echo "rocprof rc $?")
python3 - <<'PY'
import csv, os
p = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03_rocsparse_prof/t_kernel_stats.csv")
for r in csv.DictReader(open(p)):
    if float(r["AverageNs"]) > 20000:
        print(f'{r["Name"][:110]:110s} {r["Calls"]:>5s} calls  avg {float(r["AverageNs"]) / 1e3:9.1f} us')
PY
AKS_PMC_OUT=prof_banded AKS_PMC_EXTRA=1 bash profiles/collect_pmc.sh --workload banded --rows 1500000 --per-row 35 --nev 20 --max-dim 41 2>&1 | tail -60
