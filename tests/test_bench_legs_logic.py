"""CPU-only: the control flow of ``bench.py``'s pre-GPU legs (N > 1) with two ranks as two threads over real rendezvous hubs
and canned children -- which ranks start which child, how a failure on ONE rank reaches ALL ranks, the legs' time budget,
the self-validation of the legs against the default configuration, and the scaling model's use of this invocation's own
measurements.  (The legs themselves -- real children, real kernels -- run in the GPU suite:
tests/test_gpu_parity.py::test_bench_multi_rank_line_without_torch and neighbours.)"""
import socket
import sys
import threading

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402

LEGS = ["preflight", "allreduce_probe", "oneshot", "graph_replay", "torch_backend", "one_gpu_shard"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Ranks:
    """What ``_pre_gpu_legs`` needs of ``bench.Ranks``: sizes and a rendezvous hub."""

    def __init__(self, rank, world, hub):
        self.rank, self.world, self.local_rank = rank, world, rank
        self.comm = type("C", (), {"_hub": hub})()


def _run(monkeypatch, child, world=2, extra_env=None, argv=()):
    """``bench._pre_gpu_legs`` on ``world`` thread ranks; ``child(rank, leg, argv, env) -> report`` stands in for the child
    process every rank would start.  Returns ([legs of rank r], [(rank, leg, argv, env) of every child started])."""
    from arnoldi_amd.dist import _Hub

    monkeypatch.setenv("AKS_COMM_TOKEN", "legs-logic")
    for k, v in (extra_env or {}).items():
        monkeypatch.setenv(k, v)
    started, lock = [], threading.Lock()
    tls = threading.local()

    def fake_child(leg_argv, env, timeout_s):
        leg = ("one_gpu_shard" if leg_argv[leg_argv.index("--leg") + 1] == "measure" else
               {"AKS_ALLREDUCE": "oneshot", "AKS_GRAPH_COMM": "graph_replay", "AKS_HOST_ALLOC": "torch_backend"}.get(
                   next((k for k in ("AKS_ALLREDUCE", "AKS_GRAPH_COMM", "AKS_HOST_ALLOC") if k in env and leg_argv[leg_argv.index("--leg") + 1] == "solve"), None),
                   leg_argv[leg_argv.index("--leg") + 1]))
        with lock:
            started.append((tls.rank, leg, list(leg_argv), dict(env, __timeout__=timeout_s)))
        return child(tls.rank, leg, leg_argv, env)

    monkeypatch.setattr(bench, "run_own_child", fake_child)
    args = bench.parse_args(["--gpus", str(world), "--rows", "400000", *argv])
    port = _free_port()
    out, errors = [None] * world, []

    def body(r):
        tls.rank = r
        try:
            hub = _Hub(r, world, "127.0.0.1", port, 20.0)
            out[r] = bench._pre_gpu_legs(args, _Ranks(r, world, hub), hub)
            hub.close()
        except BaseException as e:              # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert not errors, errors
    return out, started


def _ok_child(rank, leg, argv, env):
    if leg == "one_gpu_shard":
        return {"restarts_per_s": 500.0, "ms_per_step": 2.0, "ms_per_step_eager_probed": 2.2, "spmv_avg_ms": 0.05, "ortho_avg_ms_per_step": 0.1, "n": 200000}
    if leg == "allreduce_probe":
        return {"slowest_rank_us_per_call": {"nccl": 21.5, "oneshot": 6.0}, "exchange_probe": {"GBs_per_peer_per_direction": 44.0}}
    return {"ok": True, "restarts_per_s": 100.0 + rank, "h_check": {"fro": 5.0, "abs_sum": 25.0, "finite": True}}


def test_every_rank_starts_every_leg_with_its_own_rendezvous(monkeypatch):
    legs, started = _run(monkeypatch, _ok_child)
    for r in (0, 1):
        assert list(legs[r]) == (LEGS if r == 0 else LEGS[:-1]), list(legs[r])      # the one-GPU shard leg is rank 0's alone
        assert all(leg["all_ranks_ok"] for leg in legs[r].values())
    by_leg = {}
    for rank, leg, argv, env in started:
        by_leg.setdefault(leg, []).append((rank, argv, env))
    assert sorted(by_leg) == sorted(LEGS) and all(len(v) == (1 if k == "one_gpu_shard" else 2) for k, v in by_leg.items())
    ports = set()
    for leg, runs in by_leg.items():
        if leg == "one_gpu_shard":
            (rank, argv, env), = runs
            assert rank == 0 and env["WORLD_SIZE"] == "1" and argv[argv.index("--rows") + 1] == "200000"      # n / N rows, one GPU
            continue
        rdv = {env["AKS_RENDEZVOUS"] for _, _, env in runs}
        assert len(rdv) == 1 and {env["RANK"] for _, _, env in runs} == {"0", "1"}          # both ranks, one fresh rendezvous per leg
        ports |= rdv
        assert all("AKS_DIST_PATH" not in env for _, _, env in runs)
    assert len(ports) == 5
    env_of = {leg: runs[0][2] for leg, runs in by_leg.items()}
    assert env_of["oneshot"]["AKS_ALLREDUCE"] == "oneshot" and env_of["torch_backend"]["AKS_HOST_ALLOC"] == "torch"
    assert env_of["graph_replay"]["AKS_GRAPH"] == "1" and env_of["graph_replay"]["AKS_GRAPH_COMM"] == "exchange"
    # the exchange probe is sized like the headline's message to one peer: n (N-1)/N (1 - exp(-5/N)) entries of 16 bytes
    probe_argv = by_leg["allreduce_probe"][0][1]
    assert int(probe_argv[probe_argv.index("--exchange-probe-bytes") + 1]) == bench.ghost_bytes_per_peer(bench.parse_args(["--gpus", "2", "--rows", "400000"]), 2)


def test_a_failure_on_one_rank_is_known_to_all(monkeypatch):
    def child(rank, leg, argv, env):
        if leg == "oneshot" and rank == 1:
            return {"error": "timed out after 240 s"}
        if leg == "preflight" and rank == 0:
            return {"ok": False, "random": {"ok": False}}
        return _ok_child(rank, leg, argv, env)

    legs, _ = _run(monkeypatch, child)
    for r in (0, 1):
        assert legs[r]["oneshot"]["all_ranks_ok"] is False and legs[r]["preflight"]["all_ranks_ok"] is False
        assert legs[r]["torch_backend"]["all_ranks_ok"] and legs[r]["graph_replay"]["all_ranks_ok"]       # a failed leg costs only itself
    assert legs[0]["oneshot"]["error"] == "failed on rank 1" and legs[1]["oneshot"]["error"].startswith("timed out")


def test_the_legs_stop_when_their_budget_is_spent(monkeypatch):
    legs, started = _run(monkeypatch, _ok_child, extra_env={"AKS_BENCH_LEGS_BUDGET_S": "0"})
    assert not started
    for r in (0, 1):
        assert "preflight" not in legs[r] and all("budget" in leg["skipped"] for leg in legs[r].values()) and len(legs[r]) == 5
    legs, started = _run(monkeypatch, _ok_child, extra_env={"AKS_BENCH_SKIP_LEGS": "graph_replay,torch_backend", "AKS_BENCH_LEGS_BUDGET_S": "600"})
    assert {leg for _, leg, _, _ in started} == {"preflight", "allreduce_probe", "oneshot", "one_gpu_shard"}
    assert {leg: env["__timeout__"] for _, leg, _, env in started} == {k: float(bench.LEG_TIMEOUT_S[k]) for k in ("preflight", "allreduce_probe", "oneshot", "one_gpu_shard")}
    # the budget is a hard cap: no child may run longer than what is left of it (rank 0's clock, the same number on all ranks)
    legs, started = _run(monkeypatch, _ok_child, extra_env={"AKS_BENCH_SKIP_LEGS": "", "AKS_BENCH_LEGS_BUDGET_S": "100"})
    assert len(started) == 11 and all(20.0 <= env["__timeout__"] <= 100.0 for _, _, _, env in started)
    per_leg = {}
    for rank, leg, _, env in started:
        per_leg.setdefault(leg, set()).add(env["__timeout__"])
    assert all(len(v) == 1 for v in per_leg.values()), per_leg


def test_legs_are_checked_against_the_default_and_feed_the_model(monkeypatch):
    legs = {"oneshot": {"h_check": {"fro": 5.0 * (1 + 3e-14), "abs_sum": 25.0, "finite": True}},
            "graph_replay": {"h_check": {"fro": 5.1, "abs_sum": 25.0, "finite": True}},
            "torch_backend": {"h_check": {"fro": float("nan"), "abs_sum": 25.0, "finite": False}},
            "allreduce_probe": {"slowest_rank_us_per_call": {"nccl": 21.5}}}
    out = bench.compare_legs_with_default(legs, {"fro": 5.0, "abs_sum": 25.0, "finite": True})
    assert out["oneshot"]["h_agrees_with_default"] and out["oneshot"]["h_vs_default_rel_diff"] < 1e-13
    assert not out["graph_replay"]["h_agrees_with_default"] and abs(out["graph_replay"]["h_vs_default_rel_diff"] - 0.02) < 1e-9
    assert not out["torch_backend"]["h_agrees_with_default"] and "h_agrees_with_default" not in out["allreduce_probe"]
    # the model: every term from this invocation's legs; without them the fall-backs are named as such
    args = bench.parse_args(["--gpus", "8"])
    res = {"exchange": {"ghost_bytes_received_per_spmv_rank0": 65_000_000, "collectives_per_arnoldi_step": 3}, "m": 20, "p": 10}
    full = bench.model_fields(res, args, 8, {"one_gpu_shard": _ok_child(0, "one_gpu_shard", [], {}), "allreduce_probe": _ok_child(0, "allreduce_probe", [], {})})
    model = full["prediction_model"]
    assert model["link_GBs_per_direction"] == 44.0 and model["link_rate_source"].startswith("this invocation")
    assert model["allreduce_us"] == 21.5 and model["one_gpu_terms"].startswith("measured")
    exch = 65e6 / (44e9 * 7) * 1e3
    step = exch + 0.15 + 2 * 21.5e-3
    assert abs(model["exchange_ms_per_spmv"] - exch) < 1e-3 and abs(model["restart_ms"] - (10 * step + (2.2 - 10 * 0.15))) < 1e-2
    assert abs(full["predicted_restarts_per_s"] - 1e3 / model["restart_ms"]) < 0.1
    bare = bench.model_fields(res, args, 8, {"one_gpu_shard": _ok_child(0, "one_gpu_shard", [], {})})["prediction_model"]
    assert bare["link_rate_source"].startswith("ASSUMED") and bare["allreduce_us_source"] == "assumed"
    assert bench.model_fields(res, args, 8, {"one_gpu_shard": {"error": "x"}}) == {} and bench.model_fields(res, args, 1, {}) == {}
