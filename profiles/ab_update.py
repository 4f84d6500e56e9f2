#!/usr/bin/env python3
"""A/B of the second-pass update kernel (aks_gs_update_norm with the DGKS test forced to fire) between builds of
libarnoldi_hip.so, interleaved rounds on one box:

    python profiles/ab_update.py n rounds LIB_A.so LIB_B.so ...

Prints the median ms per launch pair (update + its norm reduction) per panel width and build, and the TB/s of the
16 n (J + 2) bytes the pass has to move."""
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
for J in widths:
    w = basis.col(J)
    r1, r2 = ws.red(1, J + 1), ws.red(2, J + 1)
    r1.zero_(); r2.copy_(torch.randn(2 * (J + 1), dtype=torch.float64, device="cuda") * 1e-3)
    r1[2 * J] = 1.0; r2[2 * J] = 1e-4; r2[2 * J + 1] = 0.0          # sqrt(red2[J]) < eta sqrt(red1[J]): the pass runs
    fn = lambda: dev.gs_update_norm(basis, J, w, ws)
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    out[str(J)] = e0.elapsed_time(e1) / 10
print(json.dumps(out))
''' % ROOT

def main():
    n, rounds, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
    widths = os.environ.get("AB_WIDTHS", "16,20,24,28,32,36,40")
    res = {lib: {} for lib in libs}
    for _ in range(rounds):
        for lib in libs:
            env = dict(os.environ, AKS_LIB_PATH=os.path.abspath(lib))
            r = subprocess.run([sys.executable, "-c", WORKER, n, widths], capture_output=True, text=True, env=env)
            if r.returncode != 0:
                print(r.stderr[-2000:]); sys.exit(1)
            for k, v in json.loads(r.stdout.strip().splitlines()[-1]).items():
                res[lib].setdefault(k, []).append(v)
    name = lambda l: os.path.basename(os.path.dirname(l))[-12:]
    print(f"n={n}: median ms over {rounds} interleaved rounds (TB/s of 16 n (J + 2) bytes)")
    print(f"{'J':>4s} " + " ".join(f"{name(l):>20s}" for l in libs))
    for J in widths.split(","):
        cells = []
        for l in libs:
            v = sorted(res[l][J])[len(res[l][J]) // 2]
            cells.append(f"{v:9.4f} ({16 * int(n) * (int(J) + 2) / v / 1e9:5.2f})")
        print(f"{J:>4s} " + " ".join(f"{c:>20s}" for c in cells))

if __name__ == "__main__":
    main()
