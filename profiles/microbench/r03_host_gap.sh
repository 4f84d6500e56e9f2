#!/bin/bash
# host_gap_probe.py in separate processes under different host settings (one box)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_host_gap.txt; : > $out
nproc >> $out; lscpu | grep -i "model name\|numa\|socket\|thread" >> $out
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' ' >> $out; echo >> $out
run() { echo "--- $*" >> $out; env "$@" timeout -k 10 200 python profiles/host_gap_probe.py 1250000 40 >> $out 2>> gpurun_out/r03_host_gap.err || echo FAILED >> $out; tail -1 $out | cut -c1-200; }
for i in 1 2 3 4 5 6; do run X=default; done
run0() { echo "--- taskset $1" >> $out; taskset -c $1 timeout -k 10 200 python profiles/host_gap_probe.py 1250000 40 >> $out 2>> gpurun_out/r03_host_gap.err || echo FAILED >> $out; tail -1 $out | cut -c1-200; }
aff=$(python3 -c "import os; print(','.join(map(str, sorted(os.sched_getaffinity(0)))))"); echo "affinity $aff" >> $out
n0=$(python3 -c "import os; a=sorted(c for c in os.sched_getaffinity(0) if c % 128 < 64); print(','.join(map(str,a)))")
n1=$(python3 -c "import os; a=sorted(c for c in os.sched_getaffinity(0) if c % 128 >= 64); print(','.join(map(str,a)))")
for i in 1 2 3; do [ -n "$n0" ] && run0 $n0; [ -n "$n1" ] && run0 $n1; done
for i in 1 2 3; do run HSA_ENABLE_INTERRUPT=0; done
for i in 1 2 3; do run OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 MKL_NUM_THREADS=1; done
for i in 1 2 3; do run PROBE_GRAPH=1; done
