#!/usr/bin/env python3
"""Config 3's sliced SpMV alone and in the solve, ONE process, ONE box, under one rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sell_trace_u4 -- python3 profiles/r05_sell_in_solve.py gpurun_out/sell_phases_u4.json
    AKS_LIB_PATH=.../libsellu8.so rocprofv3 ... -- python3 profiles/r05_sell_in_solve.py gpurun_out/sell_phases_u8.json

The script only LAUNCHES (no events of its own: the durations are the trace's); it writes how many k_sell launches each
phase made, in order, so that profiles/r05_sell_trace_summary.py can cut the trace into phases:

  alone_fresh      x, y fresh buffers                                 (what the microbenchmark measures)
  alone_vcol       x = basis column 3, y = basis column 4             (the addresses of the solve)
  after_rewrite    x rewritten by a streaming kernel (aks_scale by 1.0) before every launch -- in the solve x is the
                   column k_finish normalised a moment ago
  after_panel      a 20-column projection (0.5 GB streamed through the caches) before every launch
  after_both       the projection, then the rewrite of x, then the launch  (the order of a step: ... k_update_proj, k_finish, SpMV)
  in_solve         KrylovSchurSolver: the 41-step expansion + 3 restarts of 16 steps (k = 20, m = 41, p = 25), bench.py's leg
"""
import json
import sys

import numpy as np

sys.path.insert(0, "arnoldi-py_amd")
from arnoldi_amd import device as dev, matrices, mem  # noqa: E402
from arnoldi_amd.engine import ArnoldiContext, CsrOperator  # noqa: E402
from arnoldi_amd.krylov_schur import KrylovSchurSolver  # noqa: E402
from arnoldi_amd.utils import arg_largest_magnitude  # noqa: E402

REPS = 40


def main(out):
    A = matrices.shell_csr(549, 549, 5, 1234)
    n = A.shape[0]
    op = CsrOperator(A)
    assert op.spmv_form == "sliced"
    ctx = ArnoldiContext(op, 41)
    rng = np.random.default_rng(0)
    V0 = (rng.standard_normal((n, 21)) + 1j * rng.standard_normal((n, 21))) / np.sqrt(2 * n)
    ctx.basis.set_cols(0, V0)
    x_fresh = mem.upload(np.ascontiguousarray(V0[:, 0]), ctx.basis.device)
    y_fresh = mem.zeros(ctx.basis.ldv, mem.c128, ctx.basis.device)
    ws = ctx.ws
    phases = []

    def phase(name, x, y, before=()):
        mem.synchronize()
        for _ in range(REPS):
            for f in before:
                f()
            op.apply(x, y, ws)
        mem.synchronize()
        phases.append([name, REPS])

    xv, yv = ctx.basis.col(3), ctx.basis.col(4)
    rewrite = lambda: dev.scale(n, xv, 1.0)                                  # noqa: E731
    panel = lambda: dev.gs_project(ctx.basis, 20, ctx.basis.col(20), ws)     # noqa: E731
    phase("warmup", x_fresh, y_fresh)
    phase("alone_fresh", x_fresh, y_fresh)
    phase("alone_vcol", xv, yv)
    phase("after_rewrite", xv, yv, (rewrite,))
    phase("after_panel", xv, yv, (panel,))
    phase("after_both", xv, yv, (panel, rewrite))
    del ctx
    np.random.seed(0)
    s = KrylovSchurSolver(op, 20, 41, 25, 1e-300, arg_largest_magnitude)
    s.ctx.use_graph = False                                   # kernel by kernel, as the probed pass of bench.py
    s.start()
    for r in range(3):
        s.contract(r)
        s.expand()
    mem.synchronize()
    phases.append(["in_solve", int(s.ctx.matvecs)])
    json.dump({"phases": phases, "n": n, "nnz": int(A.nnz), "algorithmic_bytes": int(op.algorithmic_bytes())}, open(out, "w"))


if __name__ == "__main__":
    main(sys.argv[1])
