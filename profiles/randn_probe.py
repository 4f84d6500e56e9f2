"""np.random.randn(n) against utils.legacy_randn(n) (aks_legacy_randn) on this host: time and bit-equality.
    python profiles/randn_probe.py [n]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "arnoldi-py_amd"))
import numpy as np  # noqa: E402

from arnoldi_amd import utils  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
for rep in range(3):
    np.random.seed(rep)
    t = time.perf_counter()
    a = np.random.randn(n)
    t1 = time.perf_counter() - t
    res = []
    for nt in ("1", "4", "16"):
        os.environ["AKS_PLAN_THREADS"] = nt
        np.random.seed(rep)
        t = time.perf_counter()
        b = utils.legacy_randn(n)
        res.append((nt, time.perf_counter() - t, bool(np.array_equal(a, b))))
    print(f"n = {n}: numpy {t1 * 1e3:.1f} ms; native " + ", ".join(f"{nt} thread(s) {dt * 1e3:.1f} ms (equal: {eq})" for nt, dt, eq in res))

# the whole start vector: the reference's two statements against utils.rand_normalized_vector
for rep in range(2):
    np.random.seed(rep)
    t = time.perf_counter()
    w = np.random.randn(n).astype(np.complex128)
    t_as = time.perf_counter() - t
    nr = np.linalg.norm(w)
    t_nr = time.perf_counter() - t - t_as
    w /= nr
    t_ref = time.perf_counter() - t
    os.environ["AKS_PLAN_THREADS"] = "16"
    np.random.seed(rep)
    t = time.perf_counter()
    v = utils.rand_normalized_vector(n, np.complex128)
    t_new = time.perf_counter() - t
    print(f"start vector n = {n}: reference statements {t_ref * 1e3:.1f} ms (draw + astype {t_as * 1e3:.1f}, norm {t_nr * 1e3:.1f}, "
          f"division {(t_ref - t_as - t_nr) * 1e3:.1f}); utils.rand_normalized_vector {t_new * 1e3:.1f} ms; same bits: "
          f"{bool(np.array_equal(v.view(np.uint64), w.view(np.uint64)))}")
