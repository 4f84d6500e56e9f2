python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "failed_graph_capture" 2>&1 | tail -3 | cut -c1-250
python - <<'PY'
import subprocess, sys
bad = 0
for i in range(25):
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-q", "-m", "gpu", "-x", "-k", "two_host_threads"], capture_output=True, text=True)
    ok = " passed" in r.stdout and "failed" not in r.stdout
    bad += not ok
    if not ok: print(r.stdout[-800:])
print(f"two_host_threads: {25 - bad} of 25 runs passed", flush=True)
PY
