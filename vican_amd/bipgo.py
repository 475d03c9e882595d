"""Drop-in entry points: ``bipartite_se3sync`` / ``object_bipartite_se3sync``.

Same call signature, argument meaning, output format and error behaviour as the
reference (vican/bipgo.py:353-360, 493-499): edge dict in, ``{node id: SE3}`` out,
poses are world<-node, gauge = lexicographically first camera at the identity
rotation, translations of all nodes summing to zero (CG from x0 = 0).

Everything numerical runs on the GPU through ``include/vican_hip.h``; there is no
CPU FALLBACK (a missing extension or GPU raises ``VicanError``) - a GPU-less run exists
only where the caller asks for it by name: ``device="cpu"`` (below).

Extra keyword-only arguments (defaults reproduce the reference):
    info      dict that receives solver statistics (eigenvalues per iteration,
              Lanczos steps, CG iterations, phase timings)
    group     torch.distributed process group to shard timesteps over (default:
              the world group when torch.distributed is initialised)
    verbose   print phase timings
    tight     converge the translations to relres 1e-10 with Jacobi-preconditioned CG instead of
              reproducing the reference's loosely converged scipy answer (rtol 1e-5, up to metres away
              from the solution of its own system on heavy-tailed weights); `lsqr_solver` is then only
              validated.  Off by default: it deliberately breaks parity.
    cg_stop_at  (diagnostic) run exactly this many CG iterations instead of scipy's stopping test: lets a test compare
              iterate k with the reference's iterate k (tests/test_cg_iterates.py)
    device    None (default): the current GPU.  "cpu": the same solver (vican_amd/solver.py: matrix-free block Lanczos, the
              per-node polar factors, scipy's CG recurrence) on the package's own NumPy backend (vican_amd/backend_cpu.py),
              host front-end included - what BASELINE configs[0] ("... on CPU (plumbing, no GPU)") and a box without a GPU
              run, as the reference itself runs anywhere (bipgo.py:353-490 is NumPy / SciPy).  Explicit only: it is never
              chosen for the caller, it is not the oracle, and nothing on the GPU path imports it.  Orders of magnitude
              slower than the GPU path (plain NumPy, one core), sharded over `group` like the GPU path (gloo).
"""
from __future__ import annotations

import time
import warnings
from typing import Callable, Optional

import numpy as np
import torch

from . import frontend
from ._lib import VicanError
from .geometry import SE3
from .solver import (Comm, GeneralRotationSolver, LsqrTranslationSolver, RotationSolver, TightTranslationSolver, TranslationSolver,
                     with_cooperative_fallback)

__all__ = ["bipartite_se3sync", "bipartite_se3sync_arrays", "object_bipartite_se3sync", "bipartite_so3sync", "solve_problem",
           "DisconnectedGraphWarning"]


class DisconnectedGraphWarning(UserWarning):
    """The kept edges do not connect all cameras and timesteps: poses of different components are unrelated."""


def _warn_if_disconnected(prob):
    n = frontend.count_components(prob)
    if n > 1:
        warnings.warn("the pose graph has %d connected components after filtering: the eigen-problem has %d near-null "
                      "vectors and the poses of different components are in unrelated gauges (the reference returns "
                      "such a result silently)" % (n, 3 * n), DisconnectedGraphWarning, stacklevel=3)
    return n


def _is_cpu(device):
    return device is not None and str(device).split(":")[0] == "cpu"


def _device_merge(device=None):
    """The numeric half of the front-end on the GPU (device.merge_edges); VICAN_HOST_MERGE=1 (and device="cpu") keep it in NumPy."""
    import os
    if os.environ.get("VICAN_HOST_MERGE") == "1" or _is_cpu(device):
        return None
    if not torch.cuda.is_available():
        raise VicanError("no GPU visible: vican_amd has no CPU fallback")
    from .device import merge_edges
    return merge_edges


def _shard_rows(T, world, rank):
    return (T * rank) // world, (T * (rank + 1)) // world


# Small-graph policy of multi-rank groups.  Sharding the timestep rows over N ranks removes (1 - 1/N) of the EDGE work of a
# solve and adds ~30-70 latency-bound collectives plus the launch-sequence schedule in place of the cooperative kernels; the
# camera side (Lanczos steps, Ritz, per-camera SVDs: most of a capture-sized solve) is replicated either way.  Below the
# threshold every rank solves the WHOLE graph with the single-rank schedule - no collective at all, the same bits on every
# rank (the single-rank solve is bit-reproducible), 1.0 x the one-GPU time instead of the measured ~0.5 x.  Measured on one
# MI355X (tools/shard_threshold.py, profiles/r06_shard_threshold.txt: single-rank solve of E edges against the sharded schedule on
# E / N edges + 3 us per collective for the link): at 4 M merged edges a 2 / 4 / 8-way split is predicted at 0.96-1.04 / 0.82-1.02 /
# 0.79-1.10 x, at 8 M at 1.10-1.13 / 1.26-1.27 / 1.09 x (at 25 M: 1.4 / 1.7-2.0 / 2.1-2.4 x).  VICAN_SHARD_MIN_EDGES overrides
# (0: always shard).
SHARD_MIN_EDGES = 6_000_000


def shard_policy(n_edges_total, world):
    """'sharded' or 'replicated' for a graph of n_edges_total merged edges in a group of `world` ranks."""
    import os
    if world <= 1:
        return "single"
    lim = int(os.environ.get("VICAN_SHARD_MIN_EDGES", SHARD_MIN_EDGES))
    return "sharded" if n_edges_total >= lim else "replicated"


def solve_problem(prob: frontend.Problem, maxiter: int, lsqr_solver: str, dtype=np.float32,
                  group=None, info: Optional[dict] = None, device=None, eig_tol=1e-10, tight=False, cg_stop_at=None, comm=None):
    """Solve a flattened problem on this rank's GPU; returns host arrays
    (Rc [C,3,3], Rt [T,3,3] world<-node, p_c [C,3], p_t [T,3]).  comm: a ready solver.Comm (default: Comm(group))."""
    if lsqr_solver not in ("conjugate_gradient", "direct"):
        # the reference falls through both branches and dies on the unbound result (bipgo.py:476-487)
        raise UnboundLocalError("local variable 't_est' referenced before assignment")
    if maxiter < 1:
        # the reference's loop body never runs and `r_c` is unbound at bipgo.py:346
        raise UnboundLocalError("local variable 'r_c' referenced before assignment")
    cpu = _is_cpu(device)
    if not cpu and not torch.cuda.is_available():
        raise VicanError("no GPU visible: vican_amd has no CPU fallback (device=\"cpu\" asks for the NumPy backend explicitly)")
    comm = (Comm(group, transport="torch") if cpu else Comm(group)) if comm is None else comm
    policy = shard_policy(prob.n_edges, comm.world)
    if policy == "replicated" and not getattr(comm, "force_sharded", False):
        comm = Comm.single()               # every rank solves the whole graph (small-graph policy above): no collective
    tdt = torch.float32 if np.dtype(dtype) == np.float32 else torch.float64
    T = prob.n_time
    r0, r1 = _shard_rows(T, comm.world, comm.rank)
    rp_h = prob.host_csr()[0]
    e0, e1 = int(rp_h[r0]), int(rp_h[r1])
    t0 = time.perf_counter()
    if cpu:
        return _solve_problem_cpu(prob, maxiter, lsqr_solver, dtype, info, eig_tol, tight, cg_stop_at, comm, policy, (r0, r1, e0, e1), t0)
    from .device import TILE_CAMS, download, make_backend, upload      # needs the GPU + extension
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device

    # this rank's rows of the problem: host arrays of merge_host (ONE upload through page-locked staging, device.upload) or
    # device tensors of device.merge_edges.  The diagonal of the reference's J^T J as scipy forms it (frontend.merge_host)
    # comes along; its camera part is all-reduced by the solver, so rank 0 carries it and the other ranks contribute zeros
    f64 = torch.float64
    parts = [(prob.row_ptr[r0:r1 + 1] - prob.row_ptr[r0], torch.int32), (prob.col[e0:e1], torch.int32), (prob.blk[e0:e1], tdt),
             (prob.a[e0:e1], tdt), (prob.w[e0:e1], f64), (prob.u[e0:e1], f64), (prob.v[e0:e1], f64)]
    have_deg = getattr(prob, "deg_t", None) is not None and getattr(prob, "deg_c", None) is not None
    if have_deg:
        parts += [(prob.deg_t[r0:r1], f64), (prob.deg_c, f64)]
    if any(torch.is_tensor(a) for a, _ in parts):
        parts = [a.to(dev, dt) if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt) for a, dt in parts]
    else:
        parts = upload(dev, parts)
    deg_t = deg_c = None
    if have_deg:
        deg_t = parts[7]
        deg_c = parts[8] if comm.rank == 0 else torch.zeros(prob.n_cam, dtype=f64, device=dev)
    g, K = make_backend(prob.n_cam, *parts[:7], deg_t=deg_t, deg_c=deg_c, row_ptr_host=np.asarray(rp_h[r0:r1 + 1]) - int(rp_h[r0]))
    t1 = time.perf_counter()
    nloc = r1 - r0
    bounds = [_shard_rows(T, comm.world, r)[0] for r in range(comm.world)] + [T]

    def gather_rows(loc, width):          # one all-gather of the ranks' row blocks
        return comm.gather_rows(loc.reshape(-1, width), nloc, bounds)

    # camera-tiled graphs keep their rows in an order of their own (device.TiledGraph.row_perm: packed for the shared chunking);
    # per-row results come back in the problem's order
    unperm = getattr(g, "unpermute_rows", None) or (lambda x: x)

    tm = {}

    def stages():
        rot = RotationSolver(K, comm, eig_tol=eig_tol)
        rc, Rt_loc = rot.run(maxiter)
        if info is not None:                   # phase timings wanted: costs a pipeline bubble between the two stages
            K.synchronize()
        tm["t2"] = time.perf_counter()
        # (the rotations of all rows are needed before the translation stage only by the LSQR branch - |b|^2 of the un-merged
        #  system on the host; otherwise they travel with the translations in ONE all-gather at the end)
        Rt_all = gather_rows(unperm(Rt_loc.reshape(-1, 9)[:max(nloc, 1)]), 9) if lsqr_solver == "direct" and not tight else None
        if tight:                                                            # not in the reference (module docstring)
            tr = TightTranslationSolver(K, comm)
            tr.setup(rc, Rt_loc)
            x_c, x_t = tr.solve(3 * (prob.n_cam + T))
            if not tr.info["converged"] and not K.barrier_aborted():
                raise AssertionError("tight CG did not converge")
        elif lsqr_solver == "direct":                                        # bipgo.py:479-480
            Rc_h = rc.reshape(prob.n_cam, 3, 3).transpose(1, 2).cpu().numpy()
            Rt_h = Rt_all.reshape(T, 3, 3).transpose(1, 2).cpu().numpy()
            tr = LsqrTranslationSolver(K, comm)
            x_c, x_t = tr.solve(rc, Rt_loc, 3 * (prob.n_cam + T), frontend.bnorm2(prob, Rc_h, Rt_h))
        else:                                                                # bipgo.py:476-478
            tr = TranslationSolver(K, comm)
            tr.setup(rc, Rt_loc)
            x_c, x_t = tr.solve(3 * (prob.n_cam + T), stop_at=cg_stop_at)
            if not tr.info["converged"] and not K.barrier_aborted() and cg_stop_at is None:
                raise AssertionError("CG did not converge (scipy exit_code != 0, bipgo.py:478)")
        return rot, tr, rc, Rt_all, Rt_loc, x_c, x_t

    # (a cooperative kernel whose grid barrier timed out - device shared - makes the stages run again on the launch sequences)
    rot, tr, rc, Rt_all, Rt_loc, x_c, x_t = with_cooperative_fallback(K, comm, stages)
    if comm.world > 1 and hasattr(comm, "healthy") and not getattr(comm, "_verified", False):
        # the first sharded solve of a communicator ends with ONE collective health check: if a wait of the peer exchange timed
        # out on any rank (its messages came back as NaN) the group falls back to RCCL / torch on every rank and the solve runs again
        if not comm.healthy():
            warnings.warn("vican_amd: " + "; ".join(comm.notes), RuntimeWarning, stacklevel=2)
            rot, tr, rc, Rt_all, Rt_loc, x_c, x_t = with_cooperative_fallback(K, comm, stages)
        comm._verified = True
    if hasattr(comm, "check"):
        comm.check()                       # (a timed-out wait of the peer exchange: NaN messages - an error, never a silent result)
    t2 = tm["t2"]
    t3 = time.perf_counter()
    if Rt_all is None:
        both = gather_rows(unperm(torch.cat([Rt_loc.reshape(-1, 9)[:max(nloc, 1)], x_t.reshape(-1, 3)[:max(nloc, 1)]], 1)), 12)
        Rt_all, xt_all = both[:, :9].contiguous(), both[:, 9:].contiguous()
    else:
        xt_all = gather_rows(unperm(x_t.reshape(-1, 3)[:max(nloc, 1)]), 3)
    Rc, Rt, xc_h, xt_h = download([rc.reshape(prob.n_cam, 3, 3).transpose(1, 2),          # bipgo.py:346
                                   Rt_all.reshape(T, 3, 3).transpose(1, 2), x_c, xt_all])  # bipgo.py:348
    if info is not None:
        info.update(evals=np.array(rot.stats["evals"]), lanczos_steps=list(rot.stats["lanczos_steps"]),
                    eig_resid=list(rot.stats["resid"]), sweeps=rot.stats["sweeps"], restarts=rot.stats["restarts"],
                    early_exit=rot.stats.get("early_exit"),
                    cg_iters=tr.info.get("cg_iters"), cg_relres=tr.info.get("relres"), lsqr_iters=tr.info.get("lsqr_iters"),
                    lsqr_istop=tr.info.get("istop"), n_cam=prob.n_cam, n_time=T,
                    n_edges=prob.n_edges, n_src=prob.n_src, n_chunk=getattr(g, "n_chunk", None), n_wg=getattr(g, "n_wg", None), layout=g.layout,
                    t_pack=t1 - t0, t_rot=t2 - t1, t_trans=t3 - t2, world=comm.world, policy=policy,
                    transport=getattr(comm, "transport", None), n_allreduce=getattr(comm, "n_allreduce", None))
    return Rc, Rt, xc_h, xt_h


def _solve_problem_cpu(prob, maxiter, lsqr_solver, dtype, info, eig_tol, tight, cg_stop_at, comm, policy, shard, t0):
    """solve_problem(device="cpu"): this rank's rows on the NumPy backend (vican_amd/backend_cpu.py), the same stages."""
    from .backend_cpu import NumpyBackend
    r0, r1, e0, e1 = shard
    T, nloc = prob.n_time, shard[1] - shard[0]
    host = lambda a: a.cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    rp = host(prob.row_ptr)
    deg_t = None if getattr(prob, "deg_t", None) is None else host(prob.deg_t)[r0:r1]
    deg_c = None if getattr(prob, "deg_c", None) is None else (host(prob.deg_c) if comm.rank == 0 else np.zeros_like(host(prob.deg_c)))
    K = NumpyBackend(prob.n_cam, rp[r0:r1 + 1] - rp[r0], host(prob.col)[e0:e1], host(prob.blk)[e0:e1], host(prob.a)[e0:e1], host(prob.w)[e0:e1],
                     host(prob.u)[e0:e1], host(prob.v)[e0:e1], storage=np.dtype(dtype).type, deg_t=deg_t, deg_c=deg_c)
    t1 = time.perf_counter()
    bounds = [_shard_rows(T, comm.world, r)[0] for r in range(comm.world)] + [T]
    gather = lambda loc, width: comm.gather_rows(loc.reshape(-1, width), nloc, bounds)
    rot = RotationSolver(K, comm, eig_tol=eig_tol)
    rc, Rt_loc = rot.run(maxiter)
    t2 = time.perf_counter()
    Rt_all = gather(Rt_loc.reshape(-1, 9)[:max(nloc, 1)], 9)
    if tight:
        tr = TightTranslationSolver(K, comm)
        tr.setup(rc, Rt_loc)
        x_c, x_t = tr.solve(3 * (prob.n_cam + T))
        if not tr.info["converged"]:
            raise AssertionError("tight CG did not converge")
    elif lsqr_solver == "direct":                                            # bipgo.py:479-480
        Rc_h = rc.reshape(prob.n_cam, 3, 3).transpose(1, 2).numpy()
        Rt_h = Rt_all.reshape(T, 3, 3).transpose(1, 2).numpy()
        tr = LsqrTranslationSolver(K, comm)
        x_c, x_t = tr.solve(rc, Rt_loc, 3 * (prob.n_cam + T), frontend.bnorm2(prob, Rc_h, Rt_h))
    else:                                                                    # bipgo.py:476-478
        tr = TranslationSolver(K, comm)
        tr.setup(rc, Rt_loc)
        x_c, x_t = tr.solve(3 * (prob.n_cam + T), stop_at=cg_stop_at)
        if not tr.info["converged"] and cg_stop_at is None:
            raise AssertionError("CG did not converge (scipy exit_code != 0, bipgo.py:478)")
    t3 = time.perf_counter()
    xt_all = gather(x_t.reshape(-1, 3)[:max(nloc, 1)], 3)
    if info is not None:
        info.update(evals=np.array(rot.stats["evals"]), lanczos_steps=list(rot.stats["lanczos_steps"]),
                    eig_resid=list(rot.stats["resid"]), sweeps=rot.stats["sweeps"], restarts=rot.stats["restarts"],
                    early_exit=rot.stats.get("early_exit"),
                    cg_iters=tr.info.get("cg_iters"), cg_relres=tr.info.get("relres"), lsqr_iters=tr.info.get("lsqr_iters"),
                    lsqr_istop=tr.info.get("istop"), n_cam=prob.n_cam, n_time=T, n_edges=prob.n_edges, n_src=prob.n_src, layout="numpy",
                    t_pack=t1 - t0, t_rot=t2 - t1, t_trans=t3 - t2, world=comm.world, policy=policy, transport=getattr(comm, "transport", None),
                    n_allreduce=getattr(comm, "n_allreduce", None), device="cpu")
    return (np.ascontiguousarray(rc.reshape(prob.n_cam, 3, 3).transpose(1, 2).numpy()),           # bipgo.py:346
            np.ascontiguousarray(Rt_all.reshape(T, 3, 3).transpose(1, 2).numpy()),                  # bipgo.py:348
            x_c.numpy().copy(), xt_all.numpy().copy())


def _pose_dict(prob, Rc, Rt, pc, pt, dtype):
    """{node id: SE3(R in `dtype`, t float64)} in the reference's sorted node order (bipgo.py:485-487).  Ten thousand nodes:
    the arrays are scattered into node order at once and the SE3 objects are filled attribute by attribute (same contents as
    ``SE3(R=..., t=...)``: R and t as given, a float32 4x4 beside them) - one constructor call per node took 4 us."""
    n = len(prob.tnodes)
    R = np.empty((n, 3, 3), dtype=dtype)
    t = np.empty((n, 3), dtype=np.float64)
    R[prob.tnode_of_cam], t[prob.tnode_of_cam] = Rc, pc
    R[prob.tnode_of_time], t[prob.tnode_of_time] = Rt, pt
    pose = np.zeros((n, 4, 4), dtype=np.float32)
    pose[:, :3, :3], pose[:, :3, 3], pose[:, 3, 3] = R, t, 1.0
    out = {}
    new = SE3.__new__
    # (list(array): the per-node views in one C loop - a third faster than three indexing expressions per node)
    for name, r_i, t_i, p_i in zip(prob.tnodes.tolist(), list(R), list(t), list(pose)):
        s = new(SE3)
        s._R, s._t, s._pose = r_i, t_i, p_i
        out[name] = s
    return out


def bipartite_se3sync(src_edges: dict, constraints: dict, noise_model_r: Callable, noise_model_t: Callable,
                      edge_filter: Callable, maxiter: int, lsqr_solver: str, dtype=np.float32, *,
                      info: Optional[dict] = None, group=None, verbose: bool = False, tight: bool = False,
                      cg_stop_at: Optional[int] = None, device=None) -> dict:
    """SE(3) synchronisation of static cameras and a moving marker object
    (reference bipgo.py:353-490).  See module docstring."""
    t0 = time.perf_counter()
    prob = frontend.flatten(src_edges, constraints, noise_model_r, noise_model_t, edge_filter, dtype, merge=_device_merge(device))
    _warn_if_disconnected(prob)
    t1 = time.perf_counter()
    local = {} if info is None else info
    Rc, Rt, pc, pt = solve_problem(prob, maxiter, lsqr_solver, dtype, group, local, device=device, tight=tight, cg_stop_at=cg_stop_at)
    local["t_flatten"] = t1 - t0
    out = _pose_dict(prob, Rc, Rt, pc, pt, dtype)
    if verbose:
        print("vican_amd: %d cameras, %d timesteps, %d merged edges | flatten %.3fs pack %.3fs rot %.3fs "
              "(lanczos steps %s) trans %.3fs (%s it)" % (
                  prob.n_cam, prob.n_time, prob.n_edges, t1 - t0, local["t_pack"], local["t_rot"],
                  local["lanczos_steps"], local["t_trans"], local.get("cg_iters") or local.get("lsqr_iters")))
    return out


def bipartite_se3sync_arrays(cam_ids, time_ids, marker_ids, R, t, k_r, k_t, constraints: dict, maxiter: int, lsqr_solver: str,
                             dtype=np.float32, *, info: Optional[dict] = None, group=None, tight: bool = False, device=None) -> dict:
    """``bipartite_se3sync`` for callers that hold their detections as ARRAYS (not in the reference; same mathematics,
    gauge, output dict and errors): one entry per kept source edge - camera id, timestamp, marker id (strings, the parts
    of the reference's key ``(cam, "<t>_<marker>")``), the marker's measured pose in the camera frame (R [n,3,3],
    t [n,3]) and the two weights the reference obtains from ``noise_model_r`` / ``noise_model_t`` (arrays [n]); filtering
    is the caller's (drop the entries).  Skips the edge dict and its per-edge Python callables - the part of the
    drop-in call that dominates once the solve takes milliseconds."""
    t0 = time.perf_counter()
    prob = frontend.flatten_arrays(cam_ids, time_ids, marker_ids, R, t, k_r, k_t, constraints, dtype, merge=_device_merge(device))
    _warn_if_disconnected(prob)
    t1 = time.perf_counter()
    local = {} if info is None else info
    Rc, Rt, pc, pt = solve_problem(prob, maxiter, lsqr_solver, dtype, group, local, device=device, tight=tight)
    local["t_flatten"] = t1 - t0
    return _pose_dict(prob, Rc, Rt, pc, pt, dtype)


def object_bipartite_se3sync(src_edges: dict, noise_model_r: Callable, noise_model_t: Callable,
                             edge_filter: Callable, maxiter: int, lsqr_solver: str, dtype=np.float32, *,
                             info: Optional[dict] = None, group=None, verbose: bool = False, tight: bool = False, device=None) -> dict:
    """Object (marker cube) calibration from a moving camera (reference bipgo.py:493-545):
    markers take the camera role, frames the timestep role, every pose is inverted, the
    numerically smallest marker id is pinned to the identity; only marker poses are returned."""
    t0 = time.perf_counter()
    root, prob = frontend.flatten_object(src_edges, noise_model_r, noise_model_t, edge_filter, dtype, merge=_device_merge(device))
    _warn_if_disconnected(prob)
    t1 = time.perf_counter()
    local = {} if info is None else info
    Rc, Rt, pc, pt = solve_problem(prob, maxiter, lsqr_solver, dtype, group, local, device=device, tight=tight)
    local["t_flatten"] = t1 - t0
    if verbose:
        print("vican_amd (object mode): %d markers, %d frames, %d merged edges | flatten %.3fs pack %.3fs rot %.3fs trans %.3fs" % (
            prob.n_cam, prob.n_time, prob.n_edges, t1 - t0, local["t_pack"], local["t_rot"], local["t_trans"]))
    out = _pose_dict(prob, Rc, Rt, pc, pt, dtype)
    return {k: v for k, v in out.items() if "_" not in k}               # bipgo.py:543


def bipartite_so3sync(src_edges: dict, constraints: dict, noise_model: Callable, edge_filter: Callable, maxiter: int,
                      dtype=np.float32, *, info: Optional[dict] = None, verbose: bool = False, eig_tol: float = 1e-10) -> dict:
    """SO(3) synchronisation on the bipartite graph WITHOUT eliminating the timestep nodes - drop-in for the
    reference's older variant (vican/bipgo.py:18-142; not called by its notebook).

    Returns ``{camera id: r, '<t>_0': r}`` with r the raw 3x3 blocks of the final iterate: U V^T of the last
    per-node SVD, NOT det-fixed and NOT transposed (bipgo.py:126-127,135-141).  The gauge is the first node in
    np.unique order of the 'c<id>' / 't<timestamp>' names.  Single GPU.

    ``maxiter=0`` raises the reference's UnboundLocalError.  When the dual iterate makes the connection Laplacian strongly
    indefinite the reference's shift-invert ``eigs(sigma=-1e-6)`` returns INTERIOR eigenvectors (noisy multi-marker graphs;
    the answer is a chaotic function of the input there): the solver then starts over on the dense Laplacian and reproduces it
    (``info["interior_from"]``; solver.GeneralRotationSolver._interior_step) up to 6144 unknowns, and raises an ArithmeticError
    beyond that."""
    from .device import HipBackend, LocalGraph, upload

    if not torch.cuda.is_available():
        raise VicanError("no GPU visible: vican_amd has no CPU fallback")
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        if torch.distributed.get_rank() != 0 and verbose:
            print("bipartite_so3sync: replicated on every rank (the non-eliminated variant is not sharded)")
    t0 = time.perf_counter()
    prob = frontend.flatten_so3(src_edges, constraints, noise_model, edge_filter)
    _warn_if_disconnected(prob)
    if maxiter < 1:
        raise UnboundLocalError("local variable 'r' referenced before assignment")      # bipgo.py:139
    dev = torch.device("cuda", torch.cuda.current_device())
    tdt = torch.float32 if np.dtype(dtype) == np.float32 else torch.float64
    # (the one-pass operator of the non-eliminated variant - sweep MODE 2 - exists for the block layout)
    g = LocalGraph(prob.n_cam, *upload(dev, [(prob.row_ptr, torch.int32), (prob.col, torch.int32), (prob.blk, tdt), (prob.a, tdt)]), layout="block")
    K = HipBackend(g)
    t1 = time.perf_counter()
    rot = GeneralRotationSolver(K, Comm.single(), eig_tol=eig_tol)
    rot.dense_edges = (prob.row_ptr, prob.col, prob.blk)                               # (the interior regime places R~ exactly)
    r = rot.run(maxiter).reshape(prob.n_cam + prob.n_time, 3, 3).cpu().numpy()
    t2 = time.perf_counter()
    # the reference's `r` stays in the dtype of its first eigs call unless a later iteration rebuilds it in float64
    out_dt = np.float32 if (maxiter == 1 and np.dtype(dtype) == np.float32) else np.float64
    r = r.astype(out_dt)
    out = {}
    for i, c in enumerate(prob.cam_names):                                              # bipgo.py:135-141
        out[str(c)] = r[i]
    for i, s in enumerate(prob.time_names):
        out[str(s) + "_0"] = r[prob.n_cam + i]
    if info is not None:
        info.update(evals=np.array(rot.stats["evals"]), lanczos_steps=list(rot.stats["lanczos_steps"]),
                    eig_resid=list(rot.stats["resid"]), sweeps=rot.stats["sweeps"], restarts=rot.stats["restarts"],
                    interior_from=rot.stats.get("interior_from"),
                    n_cam=prob.n_cam, n_time=prob.n_time, n_edges=prob.n_edges, n_src=prob.n_src,
                    t_pack=t1 - t0, t_rot=t2 - t1)
    if verbose:
        print("bipartite_so3sync: C=%d T=%d E=%d  pack %.1f ms  solve %.1f ms" % (
            prob.n_cam, prob.n_time, prob.n_edges, 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    return out
