#!/bin/bash
# final measurement campaign of a round (round 6: files r06_f_*): rocprofv3 stats + PMC passes (profiles/collect_pmc.sh) of the named workloads on the FINAL tree;
# the summaries land in gpurun_out/prof_<name>/pmc_summary.json (copied to profiles/pmc_summary[_<name>].json afterwards, which is
# where bench.py reads `roofline.traffic` from -- stamp-checked against the kernel source + header)
#   bash profiles/final_pmc.sh c5 markov | laplace2d banded shell laplace3d
cd $GRAFT_REPO_ROOT
for name in "$@"; do
  case $name in
    c5) args="" ;;
    markov) args="--workload markov --rows 10000000" ;;
    laplace2d) args="--workload laplace2d --rows 1000000 --nev 10 --max-dim 40" ;;
    banded) args="--workload banded --rows 1500000 --per-row 35 --nev 20 --max-dim 41" ;;
    shell) args="--workload shell --rows 1507005 --nev 20 --max-dim 41" ;;
    laplace3d) args="--workload laplace3d --rows 16000000 --nev 10 --max-dim 40" ;;
  esac
  AKS_PMC_OUT=prof_$name bash profiles/collect_pmc.sh $args > gpurun_out/final_pmc_$name.log 2>&1; echo "$name pmc rc $?"
  grep "pass " gpurun_out/final_pmc_$name.log | tr '\n' ' '; echo
  # keep what is judged, drop the raw per-pass CSVs (tens of MB)
  cp gpurun_out/prof_$name/pmc_summary.json gpurun_out/r06_f_pmc_summary_$name.json 2>/dev/null
  f=$(ls gpurun_out/prof_$name/trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_f_kernel_stats_$name.csv
  rm -rf gpurun_out/prof_$name
done
