#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded_full.py -x -q -m gpu -s -k "bench_ or two_host_threads or graphguard or never_captured or hardware_queue" > gpurun_out/r06_j9_tests.log 2>&1
rc=$?; tail -25 gpurun_out/r06_j9_tests.log; [ $rc -eq 0 ] || exit $rc
AKS_FORCE_COMM=1 timeout -k 10 900 python bench.py --gpus 1 --steps 10 --warmup 2 --no-workloads --no-cpu-baseline --no-real-leg > gpurun_out/r06_f_bench_forced_comm.json 2> gpurun_out/r06_f_bench_forced_comm.err
rc=$?; python -c "
import json;o=json.loads(open('gpurun_out/r06_f_bench_forced_comm.json').read().strip().splitlines()[-1]);print('forced comm:',o['value'],o['runtime'],o['h_check']);print({k:{a:v.get(a) for a in ('restarts_per_s','restarts_per_s_hipgraph','graphs_captured','allreduce_path','allreduce_device_us_per_call_rank0','h_agrees_with_default','h_vs_default_rel_diff','error','slowest_rank_us_per_call','seconds')} for k,v in o['legs'].items()})"
exit $rc
