"""Multi-rank path on CPU: timestep-sharded solve with world_size 2 over gloo, driven through the
NumPy stand-in backend (vican_amd/backend_cpu.py).  Checks that the sharding + all-reduce logic of
vican_amd/solver.py reproduces the single-rank result and the reference goldens, and counts the
collectives (camera-side partials only - no edge data moves)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_cases as gc
from vican_amd.backend_cpu import NumpyBackend
from test_solver_cpu import flatten_case, to_pose_arrays
from util import expected, translation_tol
from vican_amd.bipgo import _shard_rows
from vican_amd.geometry import geodesic
from vican_amd.solver import Comm, solve_on_backend


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, dt, out_q, tight=False, messages=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    if messages is None:                    # the default of sharded runs: scipy's recurrence, two messages per iteration
        os.environ.pop("VICAN_CG_MESSAGES", None)
        messages = 2
    else:
        os.environ["VICAN_CG_MESSAGES"] = str(messages)
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g, case, prob = flatten_case(name, dt)
        T = prob.n_time
        r0, r1 = _shard_rows(T, world, rank)
        e0, e1 = int(prob.row_ptr[r0]), int(prob.row_ptr[r1])
        K = NumpyBackend(prob.n_cam, prob.row_ptr[r0:r1 + 1] - prob.row_ptr[r0], prob.col[e0:e1], prob.blk[e0:e1],
                         prob.a[e0:e1], prob.w[e0:e1], prob.u[e0:e1], prob.v[e0:e1], storage=np.dtype(dt).type,
                         deg_t=prob.deg_t[r0:r1], deg_c=prob.deg_c if rank == 0 else np.zeros_like(prob.deg_c))
        comm = Comm()
        rc, Rt, x_c, x_t, stats = solve_on_backend(K, comm, gc.MAXITER, 3 * (prob.n_cam + T), tight=tight)
        full = torch.zeros(T, 12, dtype=torch.float64)
        full[r0:r1, :9] = Rt[: r1 - r0]
        full[r0:r1, 9:] = x_t[: r1 - r0]
        comm.allreduce(full)
        # the replicated camera side must be the SAME BITS on every rank (rank-ordered / fixed-slice sums behind every collective)
        import hashlib
        sums = [None] * world
        dist.all_gather_object(sums, hashlib.sha1(rc.numpy().tobytes() + x_c.numpy().tobytes()).hexdigest())
        assert len(set(sums)) == 1, "camera-side results differ between ranks: %s" % sums
        if rank == 0:
            out_q.put(dict(rc=rc.numpy().copy(), Rt=full[:, :9].numpy().copy(), x_c=x_c.numpy().copy(),
                           x_t=full[:, 9:].numpy().copy(), cg_iters=stats["cg_iters"], sweeps=stats["sweeps"],
                           n_allreduce=comm.n_allreduce, lanczos=stats["lanczos_steps"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("messages", [None, 1, 2])
@pytest.mark.parametrize("name,dt", [("g2_small", "float64"), ("g3_medium", "float64"), ("g1_object", "float32")])
def test_two_ranks_match_reference_and_single_rank(name, dt, messages):
    """messages: all-reduces per CG iteration - None: the default of sharded runs = 2: scipy's recurrence ([q_c | p.q], then
    r.r; SURVEY.md 8(e) parity mode); 1: the Chronopoulos-Gear arrangement (opt-in, VICAN_CG_MESSAGES=1)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, dt, q, False, messages)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    messages = 2 if messages is None else messages
    g, case, prob = flatten_case(name, dt)
    exp = expected(g, "conjugate_gradient", dt)
    R, t = to_pose_arrays(prob, torch.from_numpy(res["rc"]), torch.from_numpy(res["Rt"]), torch.from_numpy(res["x_c"]),
                          torch.from_numpy(res["x_t"]), exp["keys"], case["mode"] == "object")
    f64 = dt == "float64"
    assert float(geodesic(R, exp["R"]).max()) < (1e-8 if f64 else 5e-6)
    assert float(np.linalg.norm(t - exp["t"], axis=1).max()) < translation_tol(name, dt)
    assert abs(res["cg_iters"] - int(exp["cg_iters"])) <= 1
    # single-rank run of the same code: identical up to reduction order
    K1 = NumpyBackend(prob.n_cam, prob.row_ptr, prob.col, prob.blk, prob.a, prob.w, prob.u, prob.v,
                      storage=np.dtype(dt).type, deg_t=prob.deg_t, deg_c=prob.deg_c)
    rc1, Rt1, xc1, xt1, st1 = solve_on_backend(K1, Comm(), gc.MAXITER, 3 * (prob.n_cam + prob.n_time))
    # (the two runs may detect convergence of a spectral step one Lanczos step apart - the residual test is a
    #  threshold on rounding-level-different sums - so they agree to the eigen-tolerance, not to rounding)
    assert np.abs(rc1.numpy() - res["rc"]).max() < 5e-8
    assert np.abs(Rt1.numpy()[: prob.n_time] - res["Rt"]).max() < 5e-8
    # communication volume: one camera-side all-reduce per operator application, `messages` per CG step
    # ([sum w r_t | r.s | r.r] in one; or [q_c | p.q] and r.r), plus O(1) setup messages and the final gather
    expected_msgs = (res["sweeps"] - gc.MAXITER) + messages * (res["cg_iters"] + 1) + 8
    assert res["n_allreduce"] <= expected_msgs + messages * 64   # CG runs in bursts; overshoot is bounded
    if messages == 1:
        assert res["n_allreduce"] <= (res["sweeps"] - gc.MAXITER) + 8 + 64 + res["cg_iters"] + 1
    else:
        assert res["n_allreduce"] >= (res["sweeps"] - gc.MAXITER) + 2 * res["cg_iters"]      # two per iteration, really


def test_two_ranks_tight_translations():
    """Jacobi-scaled tight solve, sharded: reaches the converged solution of the reference's own system."""
    name, dt = "g3_medium", "float64"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, dt, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g, case, prob = flatten_case(name, dt)
    exp = expected(g, "conjugate_gradient", dt)
    R, t = to_pose_arrays(prob, torch.from_numpy(res["rc"]), torch.from_numpy(res["Rt"]), torch.from_numpy(res["x_c"]),
                          torch.from_numpy(res["x_t"]), exp["keys"], False)
    assert float(np.linalg.norm(t - exp["t_tight"], axis=1).max()) < 1e-7
    assert abs(t.sum(0)).max() < 1e-9                                  # the reference's gauge: translations sum to zero


def _gather_worker(rank, world, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = Comm()
        for T in (0, 1, 5, 64):
            bounds = [_shard_rows(T, world, r)[0] for r in range(world)] + [T]
            n = bounds[rank + 1] - bounds[rank]
            loc = torch.arange(bounds[rank], bounds[rank] + max(n, 1), dtype=torch.float64)[:, None] * torch.tensor([[1.0, 10.0, 100.0]])
            full = comm.gather_rows(loc, n, bounds)
            exp = torch.arange(T, dtype=torch.float64)[:, None] * torch.tensor([[1.0, 10.0, 100.0]])
            assert full.shape == (T, 3) and torch.equal(full, exp), (T, rank)
        if rank == 0:
            out_q.put("ok")
    finally:
        dist.destroy_process_group()


def test_gather_rows_is_one_all_gather_of_uneven_blocks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    assert q.get(timeout=120) == "ok"
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0


def test_shard_rows_cover_everything():
    for T in (0, 1, 7, 100, 12345):
        for w in (1, 2, 3, 8):
            b = [_shard_rows(T, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == T
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(e - s for s, e in b) - min(e - s for s, e in b) <= 1
