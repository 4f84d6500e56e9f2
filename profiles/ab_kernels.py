#!/usr/bin/env python3
"""A/B two builds of libarnoldi_hip.so on the same GPU box, interleaved rounds (guide rule 24).

    python profiles/ab_kernels.py LIB_A.so LIB_B.so [n] [rounds]

Times the Gram-Schmidt stage entry points (project / update_project / update_norm-forced) and
aks_truncate for several panel widths through the C ABI; prints median ms per (kernel, J) per build.
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import json, os, sys
sys.path.insert(0, os.path.join(%r, "arnoldi-py_amd"))
import numpy as np, torch
from arnoldi_amd import device as dev
n = int(sys.argv[1]); widths = [int(w) for w in sys.argv[2].split(",")]
m = max(widths)
basis = dev.KrylovBasis(n, m); ws = dev.Workspace(n, m)
basis.V.copy_(torch.randn(basis.V.shape, dtype=torch.complex128, device="cuda") * (1.0 / np.sqrt(n)))
out = {}
def timeit(fn, reps=5):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for J in widths:
    w = basis.col(J)
    out[f"project J={J}"] = timeit(lambda: dev.gs_project(basis, J, w, ws))
    out[f"update_project J={J}"] = timeit(lambda: dev.gs_update_project(basis, J, w, ws))
for mm, pp in ((20, 10), (40, 15), (41, 25), (80, 65), (100, 85)):
    if mm <= m:
        Q = torch.randn(mm, pp, dtype=torch.complex128, device="cuda")
        out[f"truncate m={mm} p={pp}"] = timeit(lambda: dev.truncate(basis, mm, pp, Q))
print(json.dumps(out))
''' % ROOT

def main():
    libs = sys.argv[1:3]
    n = sys.argv[3] if len(sys.argv) > 3 else "10000000"
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    widths = os.environ.get("AB_WIDTHS", "12,16,20" if int(n) >= 8_000_000 else "12,20,28,40")
    res = {lib: {} for lib in libs}
    for _ in range(rounds):
        for lib in libs:
            path, _, extra = lib.partition(":")        # "lib.so:VAR=value" adds an environment variable
            env = dict(os.environ, AKS_LIB_PATH=os.path.abspath(path))
            if extra:
                k, _, v = extra.partition("=")
                env[k] = v
            r = subprocess.run([sys.executable, "-c", WORKER, n, widths], capture_output=True, text=True, env=env)
            if r.returncode != 0:
                print(r.stderr[-2000:]); sys.exit(1)
            for k, v in json.loads(r.stdout.strip().splitlines()[-1]).items():
                res[lib].setdefault(k, []).append(v)
    keys = list(res[libs[0]])
    print(f"n={n}  median ms over {rounds} interleaved rounds")
    print(f"{'kernel':28s} " + " ".join(f"{os.path.basename(l)[-22:]:>22s}" for l in libs))
    for k in keys:
        print(f"{k:28s} " + " ".join(f"{sorted(res[l][k])[len(res[l][k])//2]:22.4f}" for l in libs))

if __name__ == "__main__":
    main()
