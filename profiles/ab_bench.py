#!/usr/bin/env python3
"""A/B library builds on whole bench.py workloads (GPU box), interleaved rounds: SpMV launch time and
restart time per build.    python profiles/ab_bench.py ROUNDS LIB_A.so LIB_B.so -- <bench args>"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sep = sys.argv.index("--")
rounds, libs, bargs = int(sys.argv[1]), sys.argv[2:sep], sys.argv[sep + 1:]
res = {l: [] for l in libs}
for _ in range(rounds):
    for l in libs:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-real-leg", "--no-workloads"] + bargs,
                           capture_output=True, text=True, env=dict(os.environ, AKS_LIB_PATH=os.path.abspath(l)))
        if r.returncode:
            print(r.stderr[-1500:]); sys.exit(1)
        d = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
        res[l].append((d["roofline"]["avg_launch_ms"], d["ms_per_step"], d["roofline"]["spmv_form"]))
for l in libs:
    sp = sorted(x[0] for x in res[l]); ms = sorted(x[1] for x in res[l])
    print(f"{os.path.basename(l):16s} spmv median {sp[len(sp)//2]:.4f} ms  restart median {ms[len(ms)//2]:.3f} ms  form {res[l][0][2]}  {' '.join(bargs)}")
