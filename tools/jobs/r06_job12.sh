#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bench_skips or bench_falls or multi_rank_line_without_torch" > gpurun_out/r06_j12_tests.log 2>&1
rc=$?; tail -6 gpurun_out/r06_j12_tests.log; [ $rc -eq 0 ] || exit $rc
python tools/compare_against_arpack.py mark:50 --nev 5 --ncv 20 --which LR 2>&1 | grep -v amdgpu.ids | tail -12
AKS_GRAPH=1 python bench.py --steps 20 --warmup 5 --probe-every 1000 --leg measure 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('10M: eager (one probed restart in 20)', o.get('restarts_per_s_eager_probed'), 'hipgraph', o.get('restarts_per_s_hipgraph'))"
