timeout -k 10 500 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "config3" -s > gpurun_out/c3_shell.log 2>&1; tail -6 gpurun_out/c3_shell.log
timeout -k 10 300 python bench.py --workload shell --rows 1507005 --nev 20 --max-dim 41 --no-workloads --no-cpu-baseline > gpurun_out/shell_bench.json 2> gpurun_out/shell_bench.err; tail -c 300 gpurun_out/shell_bench.err
timeout -k 10 300 python bench.py --workload banded --rows 1500000 --per-row 35 --nev 20 --max-dim 41 --no-workloads --no-cpu-baseline > gpurun_out/banded_bench.json 2>/dev/null
python - <<'PY'
import json
for f in ("shell","banded"):
    d=json.loads(open(f"gpurun_out/{f}_bench.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["kernel"], d.get("roofline_ortho",{}).get("frac"))
PY
