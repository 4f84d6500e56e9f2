#!/bin/bash
# round 6, job 7: what the driver runs at round end, on the final tree: the whole -m gpu suite, smoke(), the bench command
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu --durations=12 > gpurun_out/r06_j7_suite.log 2>&1
rc=$?; tail -22 gpurun_out/r06_j7_suite.log; [ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_j7_smoke.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_f_bench.json 2> gpurun_out/r06_f_bench.err
rc=$?; python - <<'PY'
import json
o=json.loads(open('gpurun_out/r06_f_bench.json').read().strip().splitlines()[-1])
print({k:o.get(k) for k in ('value','ms_per_step','runtime','calibration')}, o['roofline'], o['roofline_ortho']['frac'], o['roofline_ortho'].get('traffic_over_algorithmic'))
PY
exit $rc
