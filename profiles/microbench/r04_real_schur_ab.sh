#!/bin/bash
# A/B of the host's Schur call while the projected matrix is exactly real: dgees (AKS_REAL_SCHUR=1, shipped) against zgees
# (=0, the reference's call), whole restarts through bench.py --leg measure, interleaved, three rounds, one box
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_real_schur_ab.txt; : > $out
run() {
  label=$1; shift
  for round in 1 2 3; do
    for rs in 0 1; do
      AKS_REAL_SCHUR=$rs timeout -k 10 300 python bench.py "$@" --steps 20 --warmup 3 --leg measure > gpurun_out/rs_$rs.json 2> gpurun_out/rs_$rs.err \
        || { echo "$label rs=$rs FAILED" >> $out; tail -3 gpurun_out/rs_$rs.err >> $out; exit 1; }
      python3 - $rs "$label" $round >> $out <<'PY'
import json, sys
rs, label, rnd = sys.argv[1:4]
d = json.loads(open(f"gpurun_out/rs_{rs}.json").read().strip().splitlines()[-1])
print(f"{label:30s} round {rnd} AKS_REAL_SCHUR={rs} restarts/s {d['restarts_per_s']:8.2f} (eager+probes {d.get('restarts_per_s_eager_probed')})  ms/restart {d['ms_per_step']}")
PY
      tail -1 $out
    done
  done
}
run "laplace2d 1M k10 m40 (config 2)" --workload laplace2d --rows 1000000 --nev 10 --max-dim 40
run "laplace3d 2M k10 m40"            --workload laplace3d --rows 2000000 --nev 10 --max-dim 40
run "laplace2d 125k k10 m40"          --workload laplace2d --rows 125000 --nev 10 --max-dim 40
run "markov 1.25M k5 m20"             --workload markov --rows 1250000
run "laplace3d 16M k10 m40 (config 4)" --workload laplace3d --rows 16000000 --nev 10 --max-dim 40
