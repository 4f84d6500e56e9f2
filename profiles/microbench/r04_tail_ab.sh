#!/bin/bash
# A/B of the launch schedule of a Gram-Schmidt step: whole restarts through bench.py (--leg measure: the probed eager
# pass AND the hipGraph replay the product uses at these sizes), builds interleaved, three rounds.
#   sepgen  rounds 1-3: k_reduce / k_finish launches of their own, generic second-pass kernel
#           (build_variant.sh sepgen -DAKS_FOLD_FINISH=0 -DAKS_UPDATE_EXACT_MAX=0)
#   sep     the same launches with the exact-width second-pass kernel k_update_nc  (build_variant.sh sep -DAKS_FOLD_FINISH=0)
#   fold    SHIPPED: the second-pass kernel books the step (no k_reduce<true>, no k_finish when normalisation is deferred)
#   tail    fold + the panel kernels' last workgroup sums the partial rows itself (build_variant.sh tail -DAKS_TAIL_PANEL=1)
cd $GRAFT_REPO_ROOT
V=profiles/microbench/variants
out=gpurun_out/r04_tail_ab.txt; : > $out
run() {   # label, bench args...
  label=$1; shift
  for round in 1 2 3; do
    for lib in ${LIBS:-sepgen sep fold tail}; do
      if [ $lib = fold ]; then path=arnoldi-py_amd/arnoldi_amd/lib/libarnoldi_hip.so; else path=$V/$lib/libarnoldi_hip.so; fi
      AKS_LIB_PATH=$PWD/$path timeout -k 10 300 python bench.py "$@" --steps 20 --warmup 3 --leg measure \
          > gpurun_out/ta_$lib.json 2> gpurun_out/ta_$lib.err || { echo "$lib $label FAILED" >> $out; tail -3 gpurun_out/ta_$lib.err >> $out; exit 1; }
      python3 - $lib "$label" $round >> $out <<'PY'
import json, sys
lib, label, rnd = sys.argv[1:4]
d = json.loads(open(f"gpurun_out/ta_{lib}.json").read().strip().splitlines()[-1])
print(f"{label:28s} round {rnd} {lib:6s} restarts/s {d['restarts_per_s']:8.2f} (eager+probes {d.get('restarts_per_s_eager_probed')})  "
      f"spmv ms {d['spmv_avg_ms']}  ortho frac {d['ortho_frac']}  second {d['second_pass_fraction']}")
PY
      tail -1 $out
    done
  done
}
run "random 1.25M k5 m20"     --rows 1250000
run "laplace3d 2M k10 m40"    --workload laplace3d --rows 2000000 --nev 10 --max-dim 40
run "markov 1.25M k5 m20"     --workload markov --rows 1250000
run "random 10M k5 m20"       --rows 10000000
run "laplace3d 16M k10 m40"   --workload laplace3d --rows 16000000 --nev 10 --max-dim 40
