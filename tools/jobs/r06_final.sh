#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06_final_suite.log 2>&1
rc=$?; tail -8 gpurun_out/r06_final_suite.log; [ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_f_bench.json 2> gpurun_out/r06_f_bench.err
rc=$?; python - <<'PY'
import json
o=json.loads(open('gpurun_out/r06_f_bench.json').read().strip().splitlines()[-1])
print({k:o.get(k) for k in ('value','ms_per_step','runtime','calibration','device')}, o['roofline']['frac'], o['roofline']['traffic'], o['roofline_ortho']['frac'])
print([(w.get('name'), w.get('restarts_per_s'), w.get('spmv_frac'), w.get('error')) for w in o.get('workloads',[])], o.get('real_arithmetic',{}).get('restarts_per_s'), o.get('cpu_baseline',{}).get('value'))
PY
exit $rc
