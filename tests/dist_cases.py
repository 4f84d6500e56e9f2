"""The cases every multi-rank worker runs (tests/dist_worker.py: ranks over torch.distributed; tests/hostcomm_worker.py:
torch-free ranks over dist.HostComm): each solves the same problem as a single-process CPU oracle run and records a
verdict that tests/test_host_logic.py::check_dist_verdicts judges.  numpy / scipy / the oracle / arnoldi_amd only."""
import numpy as np
import scipy.sparse as sp


def run_cases(comm, rank, world):
    import oracle
    from arnoldi_amd import matrices, partial_schur
    from arnoldi_amd.dist import row_offsets
    from arnoldi_amd.engine import CsrOperator

    verdict = {}

    def run_case(name, A_full, nev, seed, A_arg=None, **kw):
        """Sharded solve vs the single-process oracle on the same start vector."""
        np.random.seed(seed)
        stats = {}
        Q, T, hist = partial_schur(A_full if A_arg is None else A_arg, nev, comm=comm, stats=stats, **kw)
        np.random.seed(seed)
        okw = {k: v for k, v in kw.items()}
        Qo, To, histo = oracle.krylov_schur(A_full, nev, **okw)
        _, _, rel = oracle.eig_residuals(A_full, Q, T)
        _, _, rel_o = oracle.eig_residuals(A_full, Qo, To)
        _, _, drel = stats["solver"].true_residuals()      # evaluated shard-wise on the device(s)
        verdict[name] = {
            "device_residual_err": float(np.abs(np.sort(drel) - np.sort(rel)).max()),
            "restarts_equal": bool(np.array_equal(hist.restarts, histo.restarts)),
            "matvec_hist_equal": bool(np.array_equal(hist.matvecs, histo.matvecs)),
            "eig_err": float(np.abs(np.diag(T) - np.diag(To)).max()),
            "rel_residual": float(rel.max()),
            "rel_residual_oracle": float(rel_o.max()),
            "q_shape": list(Q.shape),
            "orth_err": float(np.abs(Q.conj().T @ Q - np.eye(nev)).max()),
            "n_ghost": int(getattr(stats["solver"].op, "n_ghost", -1)),
            "restarts": int(stats["restarts"]),
            "lazy_redos": int(stats["solver"].ctx.lazy_redos),
            "second_passes": int(stats["second_passes"]),
            "collectives_per_step": int(stats["solver"].ctx.collectives_per_step()),
            "native_comm": bool(getattr(stats["solver"].op, "native_comm", False)),
        }

    LR, LM = oracle.arg_largest_real, oracle.arg_largest_magnitude
    # 1. Markov chain (README config): neighbours a few rows away -> small ghost sets
    run_case("mark50", matrices.mark(50), 5, 0, max_dim=20, stopping_criterion=1e-8, sort_function=LR)
    # 2. 2-D Laplace: halo of one grid line per side
    run_case("laplace2d", matrices.laplace2d(30, 31), 10, 0, max_dim=40, sort_function=LM)
    # 3. random CSR with planted spectrum: nearly every remote entry is needed
    n = 6000
    Ar = matrices.random_csr(n, 5, 1234, planted=(4.0, 3.7, 3.4, 3.1, 2.8, 2.5))
    run_case("random_planted", Ar, 5, 0, max_dim=20, sort_function=LM)
    # 4. each rank builds only its own rows (bench.py's construction)
    offs = row_offsets(n, world)
    rows = matrices.random_csr(n, 5, 1234, planted=(4.0, 3.7, 3.4, 3.1, 2.8, 2.5),
                               row_range=(int(offs[rank]), int(offs[rank + 1])))
    op = CsrOperator(local_rows=rows, offsets=offs, comm=comm)
    run_case("local_rows", Ar, 5, 1, A_arg=op, max_dim=20, sort_function=LM)
    # 5. block-diagonal operator: no exchange at all
    blks = [matrices.mark(12 + r) * (1.0 - 0.1 * r) for r in range(world)]  # distinct spectra
    Ab = sp.block_diag(blks, format="csr")
    offs_b = np.concatenate([[0], np.cumsum([b.shape[0] for b in blks])])
    opb = CsrOperator(Ab, offsets=offs_b, comm=comm)
    assert not opb.any_exchange
    run_case("block_diag", Ab, 2, 3, A_arg=opb, max_dim=10, stopping_criterion=1e-8, sort_function=LR)
    # 6. complex matrix values
    rng = np.random.default_rng(5)
    Ac = (sp.random(400, 400, density=0.02, random_state=np.random.RandomState(3), dtype=np.float64)
          + 1j * sp.random(400, 400, density=0.02, random_state=np.random.RandomState(4), dtype=np.float64)
          + sp.diags_array(np.linspace(1, 6, 400))).tocsr()
    run_case("complex", Ac, 3, 2, max_dim=16, stopping_criterion=1e-8, sort_function=LM)
    del rng

    # 7. explicit restarts with deflation and the naive solver, row-sharded, vs the oracle
    from arnoldi_amd.explicit_restarts import explicit_restarts_with_deflation, naive_explicit_restarts

    ex = {}
    for name, M, nev, seed, kw in (
            ("mark30", matrices.mark(30), 4, 1, dict(max_dim=30, stopping_criterion=1e-8, sort_function=LR)),
            ("planted", Ar, 3, 0, dict(max_dim=20, stopping_criterion=1e-8))):
        np.random.seed(seed)
        st = {}
        vals, vecs, hist = explicit_restarts_with_deflation(M, nev, comm=comm, stats=st, **kw)
        np.random.seed(seed)
        vo, xo, ho = oracle.explicit_restarts_with_deflation(M, nev, **kw)
        dres = st["ctx"].residual_norms(st["eigenvectors_device"], vals)     # shard-wise + all-reduce
        res = np.linalg.norm(M @ vecs - vals * vecs, axis=0)
        ex[name] = {
            "hist_equal": bool(np.array_equal(hist.restarts, ho.restarts) and np.array_equal(hist.matvecs, ho.matvecs)),
            "eig_err": float(np.abs(vals - vo).max()),
            "res_max": float(res.max()), "res_oracle_max": float(np.linalg.norm(M @ xo - vo * xo, axis=0).max()),
            "device_residual_err": float(np.abs(dres - res).max()),
            "shape": list(vecs.shape),
        }
    np.random.seed(0)
    ritz, ok, used = naive_explicit_restarts(matrices.mark(10), 10, max_restarts=5, comm=comm)
    np.random.seed(0)
    ro, oko, usedo = oracle.naive_explicit_restarts(matrices.mark(10), 10, max_restarts=5)
    ex["naive"] = {"flags_equal": bool((ok, used) == (oko, usedo)),
                   "value_err": float(np.abs(ritz.values - ro.values).max()),
                   "true_residual": float(ritz.compute_true_residuals(ritz._source)[0]),
                   "true_residual_oracle": float(ro.compute_true_residuals(matrices.mark(10))[0]),
                   "vector_shape": list(ritz.vectors.shape)}
    verdict["explicit"] = ex

    # 8. real-arithmetic mode, row-sharded: float64 ghost exchange, real-packed shards (odd local sizes)
    import real_cases as rc

    rl = {}
    for name in ("mark30_lr", "planted_odd_n", "pair_cut_at_nev5"):
        M, nev, seed, kw = rc.cases()[name]
        np.random.seed(seed)
        st = {}
        Q, T, hist = partial_schur(M, nev, comm=comm, arithmetic="real", stats=st, **kw)
        np.random.seed(seed)
        Qo, To, histo = oracle.krylov_schur(M, nev, **kw)
        _, _, rel = oracle.eig_residuals(M, Q, T)
        _, _, rel_o = oracle.eig_residuals(M, Qo, To)
        rl[name] = {"eig_err": float(rc._match(np.diag(T), np.diag(To))), "rel": float(rel.max()),
                    "rel_oracle": float(rel_o.max()), "tol": float(st["tol"]), "restarts": int(st["restarts"]),
                    "restarts_oracle": int(histo.restarts.max()), "shape": list(Q.shape),
                    "orth_err": float(np.abs(Q.conj().T @ Q - np.eye(nev)).max()),
                    "n_local": int(st["solver"].op.n_local)}
    verdict["real"] = rl

    # 9. ranks that disagree on the partition get a ValueError on EVERY rank before any data-path collective
    #    (round 1: bench.py built a different n per rank and the ranks died in gloo's all-to-all)
    mism = {}
    n_bad = 500 + 7 * rank                                     # each rank believes in a different matrix size
    offs_bad = row_offsets(n_bad, world)
    rows_bad = matrices.random_csr(n_bad, 5, 1, row_range=(int(offs_bad[rank]), int(offs_bad[rank + 1])))
    try:
        CsrOperator(local_rows=rows_bad, offsets=offs_bad, comm=comm)
        mism["size"] = "no error"
    except ValueError as e:
        mism["size"] = "ValueError: " + str(e)[:60]
    offs_ok = row_offsets(600, world)
    shift = 1 if rank == world - 1 else 0                      # one rank cuts the rows differently
    offs_shift = offs_ok.copy()
    offs_shift[1:-1] += shift
    rows_s = matrices.random_csr(600, 5, 1, row_range=(int(offs_shift[rank]), int(offs_shift[rank + 1])))
    try:
        CsrOperator(local_rows=rows_s, offsets=offs_shift, comm=comm)
        mism["offsets"] = "no error"
    except ValueError as e:
        mism["offsets"] = "ValueError: " + str(e)[:60]
    verdict["mismatch"] = mism

    return verdict
