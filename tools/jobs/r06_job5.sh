#!/bin/bash
# round 6, job 5/6: rocprofv3 stats + PMC passes of the named workloads on the final kernel source (profiles/final_pmc.sh)
mkdir -p gpurun_out
bash profiles/final_pmc.sh "$@"
ls gpurun_out | grep r06_f_ | tr '\n' ' '
