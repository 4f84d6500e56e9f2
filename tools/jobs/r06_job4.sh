#!/bin/bash
# round 6, job 4: the tests job 3 did not reach, then the measurements on the final kernel source: the driver's bench command on the
# default backend, the same on the torch backend (same box), the forced one-rank communicator (legs with the real RCCL), cold processes
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_sharded_full.py tests/test_harness.py tests/test_gpu_parity.py -x -q -m gpu -s -k "hardware_queue or carried or never_captured or breakdown or thread_comm or slab or harness or arpack or loaded_files or multi_rank_line_without_torch" > gpurun_out/r06_j4_tests.log 2>&1
rc=$?; tail -12 gpurun_out/r06_j4_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_f_bench.json 2> gpurun_out/r06_f_bench.err
rc=$?; python - <<'PY'
import json
o=json.loads(open('gpurun_out/r06_f_bench.json').read().strip().splitlines()[-1])
print({k:o.get(k) for k in ('value','ms_per_step','runtime','calibration','device')}, o['roofline']['frac'], o['roofline']['avg_launch_ms'], o['roofline_ortho']['frac'])
print([ (w.get('name'), w.get('restarts_per_s'), w.get('spmv_frac'), w.get('error')) for w in o.get('workloads',[])], o.get('real_arithmetic',{}).get('restarts_per_s'), o.get('cpu_baseline',{}).get('value'))
PY
[ $rc -eq 0 ] || { tail -5 gpurun_out/r06_f_bench.err; exit $rc; }
AKS_HOST_ALLOC=torch timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-workloads --no-cpu-baseline --no-real-leg > gpurun_out/r06_f_bench_torch_backend.json 2> gpurun_out/r06_f_bench_torch_backend.err
rc=$?; python -c "
import json;o=json.loads(open('gpurun_out/r06_f_bench_torch_backend.json').read().strip().splitlines()[-1]);print('torch backend:',o['value'],o['runtime'],o['calibration'],o['roofline']['avg_launch_ms'])"
[ $rc -eq 0 ] || { tail -5 gpurun_out/r06_f_bench_torch_backend.err; exit $rc; }
AKS_FORCE_COMM=1 timeout -k 10 900 python bench.py --gpus 1 --steps 10 --warmup 2 --no-workloads --no-cpu-baseline --no-real-leg > gpurun_out/r06_f_bench_forced_comm.json 2> gpurun_out/r06_f_bench_forced_comm.err
rc=$?; python -c "
import json;o=json.loads(open('gpurun_out/r06_f_bench_forced_comm.json').read().strip().splitlines()[-1]);print('forced comm:',o['value'],o['runtime']);print({k:{a:v.get(a) for a in ('restarts_per_s','allreduce_path','allreduce_device_us_per_call_rank0','runtime','error','slowest_rank_us_per_call','seconds')} for k,v in o['legs'].items()})"
[ $rc -eq 0 ] || { tail -5 gpurun_out/r06_f_bench_forced_comm.err; exit $rc; }
for i in 1 2; do python profiles/cold_process_probe.py; AKS_HOST_ALLOC=torch python profiles/cold_process_probe.py; done > gpurun_out/r06_cold_process.txt 2>&1
cat gpurun_out/r06_cold_process.txt
