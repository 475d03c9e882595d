#!/usr/bin/env python3
"""Benchmark of the hot path: edges/s through the primal-dual bipartite SE(3) solve.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload stress|large_shop|sparse|wide] [--scaling weak|strong]

``--gpus N`` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (one fresh child
process per GPU, before this process touches the GPU) and relays rank 0's JSON line; under
``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`` it is one of the ranks.

One "step" = one complete solve of the synthetic graph resident in HBM: rotation
stage (maxiter = 4 primal-dual iterations, each = block-Lanczos spectral step +
projections + dual updates) followed by the translation CG.  ``value`` = merged
(camera,timestep) edges x primal-dual iterations per second of WHOLE-step time,
summed over ranks (``--scaling weak``, default for stress/sparse: every rank owns ``--timesteps``
rows of one graph; ``--scaling strong``, default for large_shop: the ``--timesteps`` rows of ONE graph
are split over the ranks by vican_amd.bipgo._shard_rows; the camera side is replicated either way).  Prints ONE JSON line on rank 0 (contract in
the task statement) with two extra objects:

  roofline      dominant kernel = the fused block operator z = R~ Lambda_T^-1 R~^T x
                (``wave_sweep_kernel<.,.,0>``, or ``block_sweep_kernel<.,.,0>`` on graphs whose rows do not fit
                the wave layout).  achieved = algorithmic bytes per launch (SURVEY.md 8(d):
                E(9s+4) + 4(T+1) + 72T + 2*72C) / mean duration of its launches inside the timed steps, from HIP
                events bound to each launch's own dispatch on the launch stream (vican_set_launch_events).
  cpu_baseline  the oracle (NumPy/SciPy port of the reference's rotation loop, same
                third-party calls) timed on rank 0's host cores on a bounded sample.

Workloads (BASELINE.json configs): ``stress`` = configs[4] "1k cameras x 100k timesteps"
at visibility rho = 0.25 (25 M merged edges, 1 GB of f32 blocks per GPU: HBM-bound;
default, it is the largest single-GPU configuration) and ``large_shop`` = configs[2]
(340 cameras x 10k timesteps x 4 cams/timestep: latency-bound, wall-clock reported); ``sparse`` = a long
realistic capture (100 cameras x 2 M timesteps x 8 cams/timestep, 16 M merged edges: HBM-bound with short rows).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="stress", choices=["stress", "large_shop", "sparse", "wide"])
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak: --timesteps rows per GPU; strong: --timesteps rows in total, split over the GPUs "
                         "(default: strong for large_shop, weak otherwise)")
    ap.add_argument("--cams", type=int, default=None)
    ap.add_argument("--timesteps", type=int, default=None, help="timestep rows per GPU (weak) / in total (strong)")
    ap.add_argument("--cams-per-t", type=int, default=None)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"], help="storage type of the 3x3 blocks")
    ap.add_argument("--maxiter", type=int, default=4)
    ap.add_argument("--no-large-shop", action="store_true", help="skip the large_shop wall-clock measurement")
    ap.add_argument("--no-sparse", action="store_true", help="skip the sparse-capture measurement (detail.sparse) of the default run")
    ap.add_argument("--no-facade", action="store_true", help="skip the four-call C boundary timing (detail.facade) of the default run")
    ap.add_argument("--no-wide", action="store_true", help="skip the camera-tiled measurement (detail.wide: 4000 cameras) of the default run")
    ap.add_argument("--no-strong", action="store_true", help="multi-GPU runs: skip the strong-scaling lines (detail.strong_scaling)")
    ap.add_argument("--no-sharded-schedule", action="store_true", help="skip the one-GPU cost of the multi-GPU schedule (detail.sharded_schedule) of the default run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--block-threads", type=int, default=None, choices=[256, 512, 768, 1024])
    ap.add_argument("--n-copy", type=int, default=None)
    ap.add_argument("--cpu-sample-timesteps", type=int, default=None)
    return ap.parse_args()


def cpu_baseline(C, cpt, T_sample, maxiter, seed=0, loop=False):
    """The oracle's FULL solve - rotation loop (explicit P via SpGEMM, eigs shift-invert, LAPACK 3x3 svd) followed by
    the translation stage (incidence matrix, normal equations, scipy cg) - on a reduced-T sample of the same
    workload; single host core (ARPACK/SuperLU/LAPACK 3x3).  loop=True is the reference-shaped variant (per-node
    Python loops, what a user of the reference runs today), loop=False the vectorised one (batched numpy svd)."""
    from oracle import bipgo_oracle as orc
    from vican_amd import synth
    g = synth.make_merged_graph_torch(C, T_sample, cpt, torch.device("cpu"), torch.float64, seed=seed)
    rp = g["row_ptr"].numpy().astype(np.int64)
    time_idx = np.repeat(np.arange(T_sample), np.diff(rp))
    cam = g["col"].numpy().astype(np.int64)
    blocks = g["blk"].numpy().reshape(-1, 3, 3)
    a = g["a"].numpy()
    E = len(cam)
    # translation stage inputs: one source edge per merged edge (k_t = a, measured translation u / w, root marker)
    w, u = g["w"].numpy(), g["u"].numpy()
    eye = np.broadcast_to(np.eye(3), (E, 3, 3))
    t0 = time.perf_counter()
    Rc, Rt = orc.so3sync_arrays(C, T_sample, cam, time_idx, blocks, a, maxiter, dtype=np.float32, loop=loop)
    t1 = time.perf_counter()
    info = {}
    orc.translation_arrays(C + T_sample, cam, C + time_idx, Rc[cam], Rt[time_idx], u / w[:, None], eye, np.zeros((E, 3)),
                           np.sqrt(w), "conjugate_gradient", np.float32, info, loop)
    t2 = time.perf_counter()
    dt = t2 - t0
    return {"value": E * maxiter / dt, "unit": "edges/s", "cores": 1, "kind": "port",
            "sample": "oracle full solve = so3sync_arrays (explicit P SpGEMM + scipy eigs(k=5,sigma=-1e-6) + %s numpy svd) "
                      "+ translation_arrays (incidence matrix, J^T J, scipy cg: %s iterations), " % (
                          "per-node (reference-shaped)" if loop else "batched (vectorised)", info.get("cg_iters")) +
                      "C=%d, T=%d, %d cams/timestep, E=%d merged edges, maxiter=%d, f32, %.1f s (rotation %.1f s + translation "
                      "%.1f s) on 1 of %d host cores" % (C, T_sample, cpt, E, maxiter, dt, t1 - t0, t2 - t1, os.cpu_count()),
            "seconds": dt, "rotation_seconds": t1 - t0, "translation_seconds": t2 - t1,
            "rotation_edges_per_s": E * maxiter / (t1 - t0)}


def cpu_baselines(C, cpt, T_sample, maxiter):
    """Both variants SURVEY.md 8(d) asks for; the headline object is the vectorised one (so that the GPU/CPU ratio is
    not inflated by interpreter overhead), the reference-shaped one rides along."""
    vec = cpu_baseline(C, cpt, T_sample, maxiter, loop=False)
    try:
        ref = cpu_baseline(C, cpt, T_sample, maxiter, loop=True)
    except Exception as exc:
        ref = {"value": None, "error": repr(exc)}
    out = dict(vec)
    out["variants"] = {"vectorised": vec, "reference_shaped": ref}
    return out


def large_shop_wall_clock(args, dev, tdt, comm):
    """Wall-clock of full solves of a large_shop-sized graph (BASELINE configs[2]): warm (solver objects reused, what a
    time series of captures pays per solve), cold (graph already packed, fresh backend + solver objects: what one call
    of the drop-in pays after the host front-end), pack (CSR arrays resident in HBM -> chunked layout + graph
    constants), and the host front-end (`flatten`: edge dict -> merged CSR, vectorised NumPy + the user's callables)
    on a dict of the same size; beside them the oracle's full solve on the host (both variants)."""
    from vican_amd import frontend, synth
    from vican_amd.device import HipBackend, LocalGraph
    from vican_amd.solver import RotationSolver, TranslationSolver
    Cl, Tl2, cl = 340, 10000, 4
    gr = synth.make_merged_graph_torch(Cl, Tl2, cl, dev, tdt, seed=0)

    def pack():
        return LocalGraph(Cl, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"])

    def solve(rot, tr, K):
        rot.stats = dict(sweeps=0, lanczos_steps=[], evals=[], restarts=0, resid=[])
        tr.poll_every = 8
        rc, Rt = rot.run(args.maxiter)
        tr.setup(rc, Rt)
        tr.solve(3 * (Cl + Tl2))
        K.synchronize()

    g2 = pack()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        g2 = pack()
    torch.cuda.synchronize()
    t_pack = (time.perf_counter() - t0) / 3
    cold = []
    for _ in range(3):                                   # fresh backend + solvers each time (first one also pays module warm-up)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K2 = HipBackend(g2)
        rot2, tr2 = RotationSolver(K2, comm), TranslationSolver(K2, comm)
        solve(rot2, tr2, K2)
        cold.append(time.perf_counter() - t0)
    cold_steps = list(rot2.stats["lanczos_steps"])
    # (warm = later solves of the SAME solver object - a time series: the check positions are remembered and, on capture-sized graphs,
    #  walked down one step per solve to the smallest counts that pass; settled after at most five solves - solver.py, spectral)
    for _ in range(7):
        solve(rot2, tr2, K2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        solve(rot2, tr2, K2)
    torch.cuda.synchronize()
    res = {"ms_per_solve": (time.perf_counter() - t0) / 5 * 1e3, "cold_ms": min(cold[1:]) * 1e3, "first_call_ms": cold[0] * 1e3,
           "t_pack_ms": t_pack * 1e3, "cameras": Cl, "timesteps": Tl2, "merged_edges": g2.n_edges,
           "lanczos_steps": rot2.stats["lanczos_steps"], "cold_lanczos_steps": cold_steps, "cg_iters": tr2.info.get("cg_iters"), "solves_timed": 5,
           "schedule": "warm: the solver object remembers where its Ritz checks passed and, checking every fourth step on capture-sized graphs, "
                       "tries one step fewer per solve until a check fails; every solve ends on a passed check of the same rule (cold: a fresh "
                       "solver object, no hints)"}
    try:                                                 # host front-end on an edge dict of this size (2 markers per view)
        from vican_amd.geometry import SE3
        scene = synth.make_scene(n_cam=Cl, n_time=Tl2, n_marker=6, seed=0)
        flat = synth.make_camera_edges(scene, cpt=cl, mpv=2, sigma_r=1e-3, sigma_t=1e-3, seed=1)
        src = synth.edges_to_dict(flat, SE3)
        cons = synth.constraints_from_scene(scene, SE3)
        unit, keep = (lambda e: 1.0), (lambda e: True)
        from vican_amd.device import merge_edges
        t0 = time.perf_counter()
        prob = frontend.flatten(src, cons, unit, unit, keep, np.float32)
        res["t_flatten_host_ms"] = (time.perf_counter() - t0) * 1e3
        frontend.flatten(src, cons, unit, unit, keep, np.float32, merge=merge_edges)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frontend.flatten(src, cons, unit, unit, keep, np.float32, merge=merge_edges)       # what the drop-in call does
        torch.cuda.synchronize()
        res["t_flatten_ms"] = (time.perf_counter() - t0) * 1e3
        res["flatten_source_edges"] = int(prob.n_src)
        # the same detections handed over as arrays (bipartite_se3sync_arrays: no edge dict, no per-edge callables)
        cams = flat["cam_key"].astype(str)
        tm = np.char.partition(flat["marker_key"].astype(str), "_")
        ones = np.ones(len(cams))
        t0 = time.perf_counter()
        frontend.flatten_arrays(cams, tm[:, 0], tm[:, 2], flat["R"], flat["t"], ones, ones, cons, np.float32)
        res["t_flatten_arrays_host_ms"] = (time.perf_counter() - t0) * 1e3
        # ... what the drop-in does: string ids -> indices on the host, everything numeric on the device (vican_merge.hip)
        ix = frontend.index_edges(cams, tm[:, 0], tm[:, 2], cons)
        merge_edges(ix, flat["R"], flat["t"], ones, ones, np.float32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ix = frontend.index_edges(cams, tm[:, 0], tm[:, 2], cons)
        t1 = time.perf_counter()
        merge_edges(ix, flat["R"], flat["t"], ones, ones, np.float32)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res["t_index_ms"], res["t_merge_device_ms"] = (t1 - t0) * 1e3, (t2 - t1) * 1e3
        res["t_flatten_arrays_ms"] = (t2 - t0) * 1e3
    except Exception as exc:
        res["t_flatten_ms"] = None
        res["flatten_error"] = repr(exc)
    if not args.no_cpu_baseline:
        try:
            res["cpu_baseline"] = cpu_baselines(Cl, cl, Tl2, args.maxiter)
        except Exception as exc:
            res["cpu_baseline"] = {"value": None, "error": repr(exc)}
    return res


def facade_timing(args, dev, tdt, shapes):
    """The four-call C boundary (include/vican_hip.h: vican_plan_create / vican_solve_rot / vican_solve_trans /
    vican_plan_destroy; csrc/vican_facade.hip) driven through ctypes alone on the benchmark's graphs: ms per full solve
    (rotation loop + translation CG, the plan kept) - what a maintainer who binds only those four calls gets, next to the
    Python driver's number for the same graph."""
    import ctypes as C
    from vican_amd import _lib, synth
    lib = _lib.load()
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}
    for name, (Cn, Tn, cpt, n_solves) in shapes.items():
        gr = synth.make_merged_graph_torch(Cn, Tn, cpt, dev, tdt, seed=0)
        E = int(gr["col"].numel())
        plan = C.c_void_p()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = lib.vican_plan_create(Cn, Tn, E, _lib.STORE_F32 if tdt == torch.float32 else _lib.STORE_F64, p(gr["row_ptr"]), p(gr["col"]),
                                   p(gr["blk"]), p(gr["a"]), p(gr["w"]), p(gr["u"]), p(gr["v"]), None, None, stream, C.byref(plan))
        torch.cuda.synchronize()
        t_plan = time.perf_counter() - t0
        if rc != 0:
            out[name] = {"error": (lib.vican_last_error() or b"?").decode()}
            continue
        del gr
        torch.cuda.empty_cache()
        try:
            rcs, Rt = torch.empty(3 * Cn, 3, dtype=torch.float64, device=dev), torch.empty(Tn, 9, dtype=torch.float64, device=dev)
            x_c, x_t = torch.empty(Cn, 3, dtype=torch.float64, device=dev), torch.empty(Tn, 3, dtype=torch.float64, device=dev)
            info = _lib.SolveInfo()

            def solve():
                _lib.check(lib.vican_solve_rot(plan, args.maxiter, 1e-10, p(rcs), p(Rt), C.byref(info), stream), "vican_solve_rot")
                _lib.check(lib.vican_solve_trans(plan, p(rcs), p(Rt), 1e-5, 0, p(x_c), p(x_t), C.byref(info), stream), "vican_solve_trans")
            for _ in range(1 if E >= 2000000 else 7):            # (capture-sized graphs: the plan's check positions settle over a few solves)
                solve()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_solves):
                solve()
            torch.cuda.synchronize()
            out[name] = {"ms_per_solve": (time.perf_counter() - t0) / n_solves * 1e3, "plan_create_ms": t_plan * 1e3, "solves_timed": n_solves,
                         "sweeps": int(info.sweeps), "lanczos_steps": int(info.lanczos_steps), "cg_iters": int(info.cg_iters),
                         "cg_converged": bool(info.cg_converged), "merged_edges": E}
        finally:
            lib.vican_plan_destroy(plan)
            torch.cuda.empty_cache()
    return out


def wide_operator(args, dev, tdt, comm):
    """More cameras than the LDS-resident sweeps hold (C = 4000 > 1024): the camera-TILED path (device.TiledBackend: tiles of <= 1024
    cameras in the wave layout with a shared chunking, the operator as one launch that reads every block once - vican_tiled_op -
    where the grid is co-resident, else a rows pass and a camera pass per tile; no fused dual update).  Reported so that the regime has a number: full solves and the operator application alone,
    against the same algorithmic bytes an untiled sweep would move (E(9s+4) + 4(T+1) + 72T + 144C)."""
    from vican_amd import synth
    from vican_amd.device import make_backend
    from vican_amd.solver import RotationSolver, TranslationSolver
    Cw, Tw, cw = 4000, 100000, 250
    gr = synth.make_merged_graph_torch(Cw, Tw, cw, dev, tdt, seed=0)
    E = int(gr["col"].numel())
    g, K = make_backend(Cw, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"])
    del gr
    torch.cuda.empty_cache()
    rot, tr = RotationSolver(K, comm), TranslationSolver(K, comm)

    def solve():
        rot.stats = dict(sweeps=0, lanczos_steps=[], evals=[], restarts=0, resid=[])
        rc, Rt = rot.run(args.maxiter)
        tr.setup(rc, Rt)
        tr.solve(3 * (Cw + Tw))
        K.synchronize()
    n_warm, n_timed = getattr(args, "wide_warmup", 1), getattr(args, "wide_steps", 2)
    for _ in range(max(n_warm, 1)):
        solve()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_timed):
        solve()
    ms = (time.perf_counter() - t0) / n_timed * 1e3
    x, z = rot.X, rot.z
    # the fused launch alone (HIP events bound to the dispatch: what a kernel trace reports)
    kern_us = None
    if getattr(K, "_fused", None) is not None:
        pairs = K.tiles[0].make_launch_timers(6)
        for pr in pairs:
            K.tiles[0].time_next_sweep(pr)
            K.block_op(rot.lamT, x, z)
        torch.cuda.synchronize()
        kern_us = float(np.median([a.elapsed_time(b) for a, b in pairs[1:]])) * 1e3
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(6)]
    for a, b in ev:
        a.record()
        K.block_op(rot.lamT, x, z)
        b.record()
    torch.cuda.synchronize()
    op_ms = float(np.median([a.elapsed_time(b) for a, b in ev[1:]]))
    s_ = 4 if args.dtype == "f32" else 8
    bytes_op = E * (9 * s_ + 4) + 4 * (Tw + 1) + 72 * Tw + 144 * Cw
    res = {"workload": "wide: %d cameras x %d timesteps x %d cams/timestep, %d merged edges, camera-tiled (%d tiles of <= 1024 cameras)" % (
               Cw, Tw, cw, E, len(g.tiles)),
           "ms_per_solve": ms, "value_edges_per_s": E * args.maxiter / (ms * 1e-3), "sweeps_per_step": rot.stats["sweeps"],
           "lanczos_steps": rot.stats["lanczos_steps"], "cg_iters": tr.info.get("cg_iters"),
           "operator_ms": op_ms, "operator_bytes_algorithmic": bytes_op,
           "operator_frac": bytes_op / (op_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "tile_layouts": [t.layout for t in g.tiles],
           "fused_single_launch": getattr(K, "_fused", None) is not None, "fused_kernel_us": kern_us,
           "fused_kernel_frac": (bytes_op / (kern_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if kern_us else None,
           "padded_slots_over_edges": g.padded_slots() / max(E, 1),
           "note": "operator = all launches of one application z = P x (event pair around them); fused_single_launch: the tiles share "
                   "their chunking and vican_tiled_op reads every block once (else a rows pass and a camera pass per tile)"}
    del K, g, rot, tr
    torch.cuda.empty_cache()
    return res


def sharded_schedule(args, dev, tdt, shapes):
    """What ONE GPU can measure of the multi-GPU schedule: warm full solves of the benchmark's graphs with
    this rank holding every row, on (a) the single-rank schedule (cooperative / resident kernels), (b) the sharded schedule
    with identity collectives, (c) the sharded schedule with every all-reduce a launch of the peer exchange (the rank's own
    mailbox slot as its peer: the same kernel, granules, waits and gate as between GPUs, minus the links) and (d) with
    ncclAllReduce on a one-rank RCCL communicator held by the C library.  ms per solve, collectives per solve, the ratio to (a)."""
    from vican_amd import synth
    from vican_amd.device import HipBackend, LocalGraph
    from vican_amd.solver import Comm, RotationSolver, TranslationSolver
    out = {}
    for name, (Cn, Tn, cpt, n_solves) in shapes.items():
        gr = synth.make_merged_graph_torch(Cn, Tn, cpt, dev, tdt, seed=0)
        g = LocalGraph(Cn, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"])
        del gr
        torch.cuda.empty_cache()
        res = {"merged_edges": g.n_edges}
        for label, mk in (("single_rank", lambda: Comm.single()), ("sharded_identity", lambda: Comm.single(force_sharded=True)),
                          ("sharded_peer_exchange", lambda: Comm.single(force_sharded=True, peer=True)),
                          ("sharded_rccl_one_rank", lambda: Comm.single(force_sharded=True, native=True))):
            try:
                comm = mk()
                K = HipBackend(g)
                rot, tr = RotationSolver(K, comm), TranslationSolver(K, comm)

                def solve():
                    rot.stats = dict(sweeps=0, lanczos_steps=[], evals=[], restarts=0, resid=[])
                    tr.poll_every = 8
                    rc, Rt = rot.run(args.maxiter)
                    tr.setup(rc, Rt)
                    tr.solve(3 * (Cn + Tn))
                    K.synchronize()
                for _ in range(3 if g.n_edges >= 2000000 else 8):    # (capture-sized graphs: the check positions settle over a few solves)
                    solve()
                n0 = comm.n_allreduce
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n_solves):
                    solve()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / n_solves * 1e3
                res[label] = {"ms_per_solve": ms, "allreduces_per_solve": (comm.n_allreduce - n0) / n_solves, "cg_iters": tr.info.get("cg_iters"),
                              "lanczos_steps": list(rot.stats["lanczos_steps"]), "speculates": bool(getattr(comm, "gateable", False))}
                comm.check()
                del K, rot, tr, comm
            except Exception as exc:
                res[label] = {"error": repr(exc)[:200]}
        base = res.get("single_rank", {}).get("ms_per_solve")
        for label in ("sharded_identity", "sharded_peer_exchange", "sharded_rccl_one_rank"):
            if base and "ms_per_solve" in res.get(label, {}):
                res[label]["ratio_to_single_rank"] = res[label]["ms_per_solve"] / base
        out[name] = res
        del g
        torch.cuda.empty_cache()
    return out


def launch_ranks(args):
    """`--gpus N` without a launcher: start N fresh rank processes (this process has not touched the GPU and never
    does), relay rank 0's output, fail if any rank fails.  Never re-executes the current process.  (Profilers: put
    rocprofv3 in front of SINGLE-rank invocations only - its preloaded library initialises the GPU in the launcher.)"""
    import socket
    import subprocess
    n = args.gpus
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    ndev = torch.cuda.device_count()                     # counting devices does not initialise the GPU
    if ndev < n and "VICAN_DIST_BACKEND" not in env:
        # fewer devices than ranks: RCCL refuses two ranks on one GPU, so the ranks share devices over gloo
        # (functional run of the sharded path; the JSON line reports `devices`)
        env["VICAN_DIST_BACKEND"] = "gloo"
        print("bench.py: %d ranks on %d device(s): ranks share devices, collectives over gloo" % (n, ndev), file=sys.stderr)
    env.update(WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # supervise ALL ranks: if one dies at start-up the others would sit in init_process_group / a collective until the
    # torch timeout - on the first non-zero exit (or after the overall limit) the rest are terminated
    import threading
    out_chunks = []
    reader = threading.Thread(target=lambda: out_chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("VICAN_BENCH_TIMEOUT_S", 3600))
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad or time.time() > deadline:
            failed = "rank %s failed" % bad if bad else "timeout"
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    rcs = [p.wait() for p in procs]
    sys.stdout.write(b"".join(out_chunks).decode())
    sys.stdout.flush()
    if failed or any(rcs):
        sys.exit("bench.py: %s; rank exit codes %s" % (failed or "a rank failed", rcs))


def measure(args, dev, tdt, comm, workload, C, Tn, cpt, scaling, world, rank, steps, warmup, backend):
    """Build the synthetic graph of one workload in HBM, run `warmup` + `steps` full solves between barriers, and return
    value / ms_per_step / config / roofline / detail of it (the JSON line's fields)."""
    from vican_amd import synth
    from vican_amd.bipgo import _shard_rows, shard_policy
    from vican_amd.device import HipBackend, LocalGraph
    from vican_amd.solver import Comm, RotationSolver, TranslationSolver
    policy = "single" if world == 1 else "sharded"
    if scaling == "weak":
        T_total, r0, Tl = Tn * world, rank * Tn, Tn
    else:                                               # one graph of Tn rows, split like solve_problem does
        T_total = Tn
        policy = shard_policy(Tn * cpt, world)           # (as vican_amd.bipgo.solve_problem decides: small graphs are not sharded)
        r0, r1 = _shard_rows(Tn, world, rank) if policy != "replicated" else (0, Tn)
        Tl = r1 - r0
    group_world = world
    if policy == "replicated":
        # below the threshold every rank solves the WHOLE graph with the single-rank schedule (no collective): the job's
        # throughput is that of one solve, whatever the number of GPUs
        comm, world = Comm.single(), 1
    gr = synth.make_merged_graph_torch(C, Tl, cpt, dev, tdt, seed=0, t_offset=r0)
    g = LocalGraph(C, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"],
                   block_threads=args.block_threads, n_copy=args.n_copy)
    E_local = g.n_edges
    del gr
    torch.cuda.empty_cache()

    class TimedBackend(HipBackend):
        """HIP events bound to every launch of the dominant kernel (vican_set_launch_events: start / stop of the
        dispatch itself on the launch stream - no event command sits in the stream next to the kernel, so the
        5-8 us of queue idle that an event pair recorded around a launch adds are not part of the measurement)."""
        events, timers = [], []
        record = False

        def block_op_raw(self, lamT_inv, x):
            if self.record and self.timers:
                pair = self.timers.pop()
                self.time_next_sweep(pair)
                self.events.append(pair)
            return super().block_op_raw(lamT_inv, x)

        def block_op(self, lamT_inv, x, z_out):
            if not self.record:
                return super().block_op(lamT_inv, x, z_out)
            self.block_op_raw(lamT_inv, x)                 # the sweep kernel alone ...
            self.fold_z(z_out)                             # ... then the slab fold

        # every OTHER heavy edge kernel (detail.kernels): events bound to the kernel's own dispatch inside one extra,
        # untimed, instrumented solve after the timed region (`profile` = dict label -> list of event pairs)
        profile = None

        def _bind(self, label):
            if self.profile is not None and self.timers:
                pair = self.timers.pop()
                self.time_next_sweep(pair)
                self.profile.setdefault(label, []).append(pair)

        def dual_update(self, *a):
            self._bind("dual_update_sweep")
            return super().dual_update(*a)

        def dual_update_op(self, *a):
            self._bind("dual_update_op_sweep")
            return super().dual_update_op(*a)

        def trans_rhs(self, *a):
            self._bind("trans_rhs")
            return super().trans_rhs(*a)

        def cg_iter_local(self, *a):
            self._bind("cg_sweep")
            return super().cg_iter_local(*a)

        def cg_iter_fused(self, *a, **k):                   # (single rank: the events land on the call's sweep launch)
            self._bind("cg_sweep")
            return super().cg_iter_fused(*a, **k)

        def lsqr_step(self, *a):
            self._bind("lsqr_step")
            return super().lsqr_step(*a)

    K = TimedBackend(g)
    K.events, K.timers = [], []
    rot = RotationSolver(K, comm)
    tr = TranslationSolver(K, comm)
    n_unknowns = 3 * (C + T_total)

    def step(split=False):
        """One full solve.  split=True also syncs between the rotation and the translation stage to time them
        separately (costs a ~40 us pipeline bubble: done in the instrumented step only)."""
        rot.stats = dict(sweeps=0, lanczos_steps=[], evals=[], restarts=0, resid=[])
        tr.poll_every = 8
        t0 = time.perf_counter()
        rc, Rt = rot.run(args.maxiter)
        if split:
            K.synchronize()
        t1 = time.perf_counter()
        tr.setup(rc, Rt)
        tr.solve(n_unknowns)
        K.synchronize()
        return t1 - t0, time.perf_counter() - t1

    def barrier():
        torch.cuda.synchronize()
        if group_world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    n_ar0 = comm.n_allreduce
    # the FIRST solve of this graph (graph packed, backend + solver objects fresh: no schedule hints, lazy allocations and
    # kernel attributes still to be set) - what one call pays after the pack; always run, counted as the first warm-up step
    barrier()
    t0 = time.perf_counter()
    step()
    first_solve_ms = (time.perf_counter() - t0) * 1e3
    for _ in range(max(warmup - 1, 0)):
        step()
    # the transport of the all-reduces is settled BEFORE the timed region: if a wait of the peer exchange timed out on any rank
    # during the warm-up (messages came back as NaN) the group falls back to RCCL / torch on every rank and warms up again
    if hasattr(comm, "healthy") and not comm.healthy():
        print("bench.py: %s" % "; ".join(comm.notes), file=sys.stderr)
        rot, tr = RotationSolver(K, comm), TranslationSolver(K, comm)
        for _ in range(max(warmup, 1)):
            step()
    K.timers = K.make_launch_timers(64)
    barrier()
    # HIP events are bound to every launch of the dominant kernel of the LAST timed step (its sync between the two
    # stages, for the stage split, costs a pipeline bubble - one instrumented step is enough).
    t_rot = t_tr = 0.0
    t0 = time.perf_counter()
    for i in range(steps):
        K.record = (i == steps - 1)
        a, b = step(split=K.record)
        if K.record:
            t_rot, t_tr = a, b
    K.record = False
    barrier()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if group_world > 1:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(el.item())
    if hasattr(comm, "check"):
        comm.check()                                     # (a timed-out wait of the peer exchange is an error, never a number)
    if K.barrier_aborted() or K.coop_failures:
        print("bench.py: a cooperative kernel's grid barrier gave up during the run (%s): device shared?" % K.coop_failures, file=sys.stderr)

    # ---- outside the timed region: a COLD solve (fresh solver objects on the warm backend: no remembered step counts, no
    # remembered CG iteration count - every convergence check that fails is paid for) and one instrumented solve with events
    # on every heavy edge kernel; a short LSQR run (lsqr_solver="direct") for its fused pass
    cold_solve_ms, kernels = None, {}
    if group_world == 1:
        rot_w, tr_w = rot, tr
        rot, tr = RotationSolver(K, comm), TranslationSolver(K, comm)
        barrier()
        t0c = time.perf_counter()
        step()
        cold_solve_ms = (time.perf_counter() - t0c) * 1e3
        cold_steps = list(rot.stats["lanczos_steps"])
        rot, tr = rot_w, tr_w
        # FIVE instrumented solves: the per-kernel figures are the MEDIAN of the five per-solve averages, with their range (the
        # same code scatters by 10-20 % between runs and boxes: a best case is not a measurement)
        s_ = 4 if args.dtype == "f32" else 8
        E_, T_ = E_local, Tl
        # algorithmic bytes per launch (SURVEY.md 8(d) / DESIGN.md section 5; V = vector passes are inside the formulas)
        kb = {"dual_update_sweep": E_ * (9 * s_ + 4) + 4 * (T_ + 1) + 72 * C + 72 * T_,
              "dual_update_op_sweep": E_ * (9 * s_ + 4) + 4 * (T_ + 1) + 144 * C + 72 * T_,
              "trans_rhs": E_ * (48 + 4) + 72 * T_ + 72 * C,
              "cg_sweep": 12 * E_ + 4 * (T_ + 1) + 96 * T_ + 48 * C,
              "lsqr_step": E_ * (12 + 48) + 48 * T_ + 48 * C}
        # what bounds each kernel: the edge streams are HBM-bound; the CG product is not - 6 LDS 64-bit
        # atomics + 3 gathers + 12 fixed-point conversions per 6-byte edge: LDS array / VALU issue (its HBM fractions are
        # reported, on the algorithmic bytes AND on the counter traffic, but they are not what limits it)
        bound = {"dual_update_sweep": "hbm", "dual_update_op_sweep": "hbm", "trans_rhs": "hbm", "cg_sweep": "lds", "lsqr_step": "hbm"}
        pmc_file = {"dual_update_op_sweep": "dual_update_op_counters", "trans_rhs": "rhs_counters", "cg_sweep": "cg_counters"}
        reps = {}
        for rep in range(5):
            K.timers = K.make_launch_timers(96)
            K.profile = {}
            step()
            if rep == 0:
                try:
                    from vican_amd.solver import LsqrTranslationSolver
                    ls = LsqrTranslationSolver(K, comm)
                    ls.solve(rot.rc, rot.Rt, n_unknowns, None, iter_lim=6)
                except Exception as exc:                              # (graphs without an LSQR layout: the figure is optional)
                    kernels["lsqr_step"] = {"error": repr(exc)[:120]}
            K.synchronize()
            prof, K.profile = K.profile, None
            for label, pairs in prof.items():
                ms = np.array([a.elapsed_time(b) for a, b in pairs])
                ms = ms[ms >= 0.1 * np.median(ms)]                     # (cancelled speculative launches exit at once)
                reps.setdefault(label, []).append((float(ms.mean() * 1e3), int(len(ms))))
        import glob as _glob
        for label, rr in reps.items():
            us = np.array([u for u, _ in rr])
            med = float(np.median(us))
            ent = {"launches": rr[0][1], "avg_us": med, "avg_us_min": float(us.min()), "avg_us_max": float(us.max()), "solves": len(rr),
                   "bytes_per_launch": int(kb[label]), "achieved_GBps": float(kb[label] / (med * 1e-6) / 1e9),
                   "frac": float(kb[label] / (med * 1e-6) / 1e9 / HBM_PEAK_GBS), "bound": bound[label],
                   "traffic_bytes": None, "hbm_frac_traffic": None, "traffic_source": None}
            # HBM bytes the kernel really moved, from the committed rocprofv3 PMC passes of this workload (null if none matches)
            for f in sorted(_glob.glob(os.path.join(ROOT, "profiles", "*_%s.json" % pmc_file.get(label, "-none-"))), reverse=True):
                try:
                    pj = json.load(open(f))
                    if pj.get("traffic") and pj.get("bench_detail", {}).get("bytes_per_launch") == kb[label]:
                        ent["traffic_bytes"] = float(pj["traffic"]["hbm_bytes"])
                        ent["hbm_frac_traffic"] = float(ent["traffic_bytes"] / (med * 1e-6) / 1e9 / HBM_PEAK_GBS)
                        ent["traffic_source"] = "profiles/" + os.path.basename(f)
                        break
                except Exception:
                    pass
            kernels[label] = ent
    kern_ms = np.array([a.elapsed_time(b) for a, b in K.events])
    # a speculative launch that the Ritz gate cancelled on the device exits at its first instruction (a few
    # microseconds): not a sweep, so not part of the average (none occur once the step prediction has settled)
    n_cancelled = int((kern_ms < 0.1 * np.median(kern_ms)).sum()) if len(kern_ms) else 0
    if n_cancelled:
        kern_ms = kern_ms[kern_ms >= 0.1 * np.median(kern_ms)]
    op_bytes = g.op_bytes()
    achieved = op_bytes / (kern_ms.mean() * 1e-3) / 1e9 if len(kern_ms) else 0.0
    n_allreduce_total = comm.n_allreduce - n_ar0
    cnt = torch.tensor([float(E_local)], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(cnt)
    E_total = int(cnt[0].item())
    value = E_total * args.maxiter * steps / elapsed
    # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes of THIS workload
    # (profiles/<tag>_sweep_counters.json, written by tools/collect_profiles.py); null if none matches
    traffic, traffic_source = None, None
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters.json")), reverse=True):
        try:
            pj = json.load(open(f))
            if pj.get("traffic") and pj.get("bytes_per_launch_algorithmic") == op_bytes:
                traffic, traffic_source = pj["traffic"]["hbm_bytes"], "profiles/" + os.path.basename(f)
                break
        except Exception:
            pass
    res = {
        "value": value, "ms_per_step": elapsed / steps * 1e3,
        "config": {"workload": "%s: %d cameras x %d timesteps (%s) x %d cams/timestep, %d merged edges in total "
                               "(%d on rank 0), maxiter=%d + CG translation solve, blocks stored %s" % (
                                   workload, C, T_total, "%d per GPU" % Tn if scaling == "weak" else "split over the GPUs",
                                   cpt, E_total, E_local, args.maxiter, args.dtype),
                   "arithmetic": "%s block products, exact 64-bit fixed-point accumulation (double-word in the translation stage), "
                                 "f64 camera side and CG" % args.dtype,
                   "parallelism": ("timestep-sharded x%d, camera side replicated" % world) if policy != "replicated" else
                                  ("replicated x%d: %d merged edges are below the sharding threshold (vican_amd.bipgo.SHARD_MIN_EDGES), every rank "
                                   "solves the whole graph with the single-rank schedule" % (group_world, Tn * cpt)),
                   "policy": policy, "transport": getattr(comm, "transport", None), "comm_notes": getattr(comm, "notes", None),
                   "devices": torch.cuda.device_count(), "dist_backend": backend if group_world > 1 else None},
        "roofline": {"bound": "hbm", "kernel": "%s_sweep_kernel<MODE=0> (vican_block_op)" % g.layout,
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_source, "bytes_per_launch": op_bytes, "launches": int(len(kern_ms)), "cancelled_speculative_launches": n_cancelled,
                     # achieved / frac are on the ALGORITHMIC bytes of SURVEY.md 8(d), E(9s + 4) + ...; wave-layout graphs carry a 2-byte
                     # index (vican_graph_t.idx16), so the sweep streams 2 bytes per edge less than that figure
                     "index_bytes_per_edge": 2 if g.layout == "wave" else 4,
                     "bytes_streamed_per_launch": int(op_bytes - (2 * E_local if g.layout == "wave" else 0)),
                     "avg_launch_ms": float(kern_ms.mean()) if len(kern_ms) else None,
                     "padded_slots_over_edges": g.padded_slots() / max(E_local, 1)},
        "detail": {"rot_loop_ms_per_step": t_rot * 1e3, "cg_ms_per_step": t_tr * 1e3,     # split of the last (instrumented) step
                   "sweeps_per_step": rot.stats["sweeps"], "lanczos_steps": rot.stats["lanczos_steps"],
                   "eig_resid": rot.stats["resid"], "cg_iters": tr.info.get("cg_iters"),
                   "lanczos_checks": rot.stats.get("n_check"),
                   "cg_converged": tr.info.get("converged"), "n_chunk": g.n_chunk, "n_wg": g.n_wg,
                   "layout": g.layout, "block_threads": g.block_threads, "n_copy": g.n_copy, "max_rows": g.max_rows,
                   "rot_edges_per_s": E_total * args.maxiter / t_rot if t_rot else None,
                   "edges_rank0": E_local, "rows_rank0": Tl,
                   "n_allreduce_per_solve": n_allreduce_total / max(steps + max(warmup, 1), 1),
                   # first_solve_ms: the very first solve on a fresh backend (lazy allocations included); cold_solve_ms: fresh
                   # solver objects on the warm backend (no schedule hints); ms_per_step above: warm re-solves (time series)
                   "first_solve_ms": first_solve_ms, "cold_solve_ms": cold_solve_ms,
                   "cold_lanczos_steps": cold_steps if group_world == 1 else None,
                   # the other heavy edge kernels: HIP events on the kernel's own dispatch in one instrumented solve after
                   # the timed region; bytes = the algorithmic formulas of DESIGN.md section 5
                   "kernels": kernels},
    }
    del K, g, rot, tr
    torch.cuda.empty_cache()
    return res


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N` or under "
                 "torch.distributed.run with --nproc-per-node N)" % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback)")
    # one process per GPU; VICAN_DIST_BACKEND=gloo lets several ranks share one device (functional test of
    # the sharded path on a 1-GPU box - RCCL refuses two ranks on the same GPU)
    backend = os.environ.get("VICAN_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if local_rank < ndev else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
    from vican_amd.solver import Comm
    if args.workload == "wide":
        # more cameras than the LDS-resident sweeps hold: the camera-tiled path as the main measurement (one GPU)
        if world != 1:
            sys.exit("bench.py: --workload wide is a single-GPU measurement")
        args.wide_steps, args.wide_warmup = args.steps, args.warmup
        tdt = torch.float32 if args.dtype == "f32" else torch.float64
        w = wide_operator(args, dev, tdt, Comm())
        # HBM traffic of the fused launch from the committed rocprofv3 PMC passes of THIS workload (profiles/<tag>_wide_counters.json)
        w_traffic, w_source = None, None
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_wide_counters.json")), reverse=True):
            try:
                pj = json.load(open(f))
                if pj.get("traffic") and pj.get("bytes_per_launch_algorithmic") == w["operator_bytes_algorithmic"]:
                    w_traffic, w_source = pj["traffic"]["hbm_bytes"], "profiles/" + os.path.basename(f)
                    break
            except Exception:
                pass
        line = {"metric": "edges/sec through bipartite_se3sync primal-dual iter", "value": w["value_edges_per_s"], "unit": "edges/s",
                "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": w["ms_per_solve"], "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": w["workload"] + ", maxiter=%d + CG translation solve, blocks stored %s" % (args.maxiter, args.dtype)},
                "roofline": {"bound": "hbm", "kernel": "tiled_sweep_kernel (vican_tiled_op)" if w["fused_single_launch"] else "wave_sweep_kernel<MODE=1|4> per tile",
                             "achieved": w["operator_bytes_algorithmic"] / ((w["fused_kernel_us"] or w["operator_ms"] * 1e3) * 1e-6) / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": w["fused_kernel_frac"] or w["operator_frac"], "traffic": w_traffic, "traffic_source": w_source,
                             "bytes_per_launch": w["operator_bytes_algorithmic"], "avg_launch_ms": (w["fused_kernel_us"] or w["operator_ms"] * 1e3) * 1e-3,
                             "padded_slots_over_edges": w["padded_slots_over_edges"]},
                "cpu_baseline": {"value": None, "unit": "edges/s", "cores": 1, "kind": "port",
                                 "sample": "not timed for this workload (the oracle forms the 12000 x 12000 power-graph matrix explicitly); "
                                           "the default command carries the CPU baseline"},
                "detail": w}
        print(json.dumps(line), flush=True)
        return
    if args.workload == "stress":
        C, Tn, cpt = args.cams or 1000, args.timesteps or 100000, args.cams_per_t or 250
    elif args.workload == "sparse":
        C, Tn, cpt = args.cams or 100, args.timesteps or 2000000, args.cams_per_t or 8
    else:
        C, Tn, cpt = args.cams or 340, args.timesteps or 10000, args.cams_per_t or 4
    scaling = args.scaling or ("strong" if args.workload == "large_shop" else "weak")
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    comm = Comm()
    m = measure(args, dev, tdt, comm, args.workload, C, Tn, cpt, scaling, world, rank, args.steps, args.warmup, backend)
    out = {
        "metric": "edges/sec through bipartite_se3sync primal-dual iter",
        "value": m["value"], "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": m["ms_per_step"], "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": m["config"], "roofline": m["roofline"], "detail": m["detail"],
    }
    if world > 1 and args.workload == "stress" and not args.no_strong:
        # beside the weak-scaling headline: STRONG scaling on a graph where sharding can pay (the stress graph's rows split over
        # the ranks) and on large_shop (40 000 merged edges: below the sharding threshold, every rank solves it whole - the
        # job's wall-clock is one GPU's, not the ~2x longer sharded one).  Every rank takes part; rank 0 reports.
        strong = {}
        for name, (Cs, Ts, cs) in (("stress_rows_split", (C, Tn, cpt)), ("large_shop", (340, 10000, 4))):
            try:
                ms_ = measure(args, dev, tdt, comm, name, Cs, Ts, cs, "strong", world, rank, max(args.steps // 2, 3), 2, backend)
                strong[name] = {"value_edges_per_s": ms_["value"], "ms_per_solve": ms_["ms_per_step"], "workload": ms_["config"]["workload"],
                                "policy": ms_["config"]["policy"], "parallelism": ms_["config"]["parallelism"],
                                "transport": ms_["config"]["transport"], "n_allreduce_per_solve": ms_["detail"]["n_allreduce_per_solve"],
                                "edges_rank0": ms_["detail"]["edges_rank0"], "rows_rank0": ms_["detail"]["rows_rank0"],
                                "cg_iters": ms_["detail"]["cg_iters"], "lanczos_steps": ms_["detail"]["lanczos_steps"],
                                "roofline_frac": ms_["roofline"]["frac"], "scaling": "strong"}
            except Exception as exc:
                strong[name] = {"error": repr(exc)[:300]}
        out["detail"]["strong_scaling"] = strong
    if rank == 0 and world == 1 and args.workload == "stress" and not args.no_large_shop:
        # second half of BASELINE.json's metric: wall-clock of a full solve of a large_shop-sized graph
        # (340 cameras x 10 000 timesteps x 4 cameras per timestep: cache-resident, latency-bound)
        try:
            out["detail"]["large_shop_wall_clock"] = large_shop_wall_clock(args, dev, tdt, comm)
        except Exception as exc:
            out["detail"]["large_shop_wall_clock"] = {"error": repr(exc)}
    if rank == 0 and world == 1 and args.workload == "stress" and not args.no_sparse:
        # the regime real captures are in (2-8 cameras per timestep): a long sparse capture, 100 cameras x 2 M timesteps x
        # 8 cameras per timestep = 16 M merged edges, the same measurement in short form (`--workload sparse` for the full one)
        try:
            ms = measure(args, dev, tdt, comm, "sparse", 100, 2000000, 8, "weak", 1, 0, 3, 1, backend)
            out["detail"]["sparse"] = {"workload": ms["config"]["workload"], "frac": ms["roofline"]["frac"],
                                       "achieved_GBps": ms["roofline"]["achieved"], "avg_launch_ms": ms["roofline"]["avg_launch_ms"],
                                       "bytes_per_launch": ms["roofline"]["bytes_per_launch"], "traffic": ms["roofline"]["traffic"],
                                       "ms_per_solve": ms["ms_per_step"], "value_edges_per_s": ms["value"],
                                       "sweeps_per_step": ms["detail"]["sweeps_per_step"], "cg_iters": ms["detail"]["cg_iters"],
                                       "cg_ms_per_step": ms["detail"]["cg_ms_per_step"], "layout": ms["detail"]["layout"],
                                       "max_rows": ms["detail"]["max_rows"], "n_copy": ms["detail"]["n_copy"], "steps": 3, "warmup": 1}
        except Exception as exc:
            out["detail"]["sparse"] = {"error": repr(exc)}
    if rank == 0 and world == 1 and args.workload == "stress" and not args.no_wide:
        try:
            out["detail"]["wide"] = wide_operator(args, dev, tdt, comm)
        except Exception as exc:
            out["detail"]["wide"] = {"error": repr(exc)[:300]}
    if rank == 0 and world == 1 and args.workload == "stress" and not args.no_facade:
        # the four-call C boundary on the same graphs, next to the Python driver's numbers
        try:
            shapes = {"stress": (C, Tn, cpt, 5), "large_shop": (340, 10000, 4, 10)}
            if not args.no_wide:
                shapes["wide"] = (4000, 100000, 250, 2)          # camera tiles behind the same four calls (csrc/vican_facade_tiles.hip)
            f = facade_timing(args, dev, tdt, shapes)
            wd = out["detail"].get("wide", {})
            if "ms_per_solve" in f.get("wide", {}) and "ms_per_solve" in wd:
                f["wide"]["python_driver_ms_per_solve"] = wd["ms_per_solve"]
                f["wide"]["ratio_to_python_driver"] = f["wide"]["ms_per_solve"] / wd["ms_per_solve"]
            if "ms_per_solve" in f.get("stress", {}):
                f["stress"]["python_driver_ms_per_solve"] = m["ms_per_step"]
                f["stress"]["ratio_to_python_driver"] = f["stress"]["ms_per_solve"] / m["ms_per_step"]
            ls = out["detail"].get("large_shop_wall_clock", {})
            if "ms_per_solve" in f.get("large_shop", {}) and "ms_per_solve" in ls:
                f["large_shop"]["python_driver_ms_per_solve"] = ls["ms_per_solve"]
                f["large_shop"]["ratio_to_python_driver"] = f["large_shop"]["ms_per_solve"] / ls["ms_per_solve"]
            out["detail"]["facade"] = f
        except Exception as exc:
            out["detail"]["facade"] = {"error": repr(exc)[:300]}
    if rank == 0 and world == 1 and args.workload == "stress" and not args.no_sharded_schedule:
        try:
            out["detail"]["sharded_schedule"] = sharded_schedule(args, dev, tdt, {"stress": (C, Tn, cpt, 5), "large_shop": (340, 10000, 4, 10)})
        except Exception as exc:
            out["detail"]["sharded_schedule"] = {"error": repr(exc)[:300]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        Ts = args.cpu_sample_timesteps or {"stress": 300, "sparse": 20000}.get(args.workload, 10000)
        try:
            out["cpu_baseline"] = cpu_baselines(C, cpt, Ts, args.maxiter)
        except Exception as exc:                                  # the baseline must never sink the line
            out["cpu_baseline"] = {"value": None, "error": repr(exc)}
        # the one SAME-SIZE CPU / GPU pair of the line (large_shop: the oracle solves the whole 340 x 10 000 graph in seconds),
        # beside the headline's reduced-T sample
        try:
            ls = out["detail"].get("large_shop_wall_clock", {})
            cb = ls.get("cpu_baseline", {})
            if cb.get("value") and "ms_per_solve" in ls:
                out["cpu_baseline"].setdefault("variants", {})["large_shop_same_size"] = {
                    "cpu_edges_per_s": cb["value"], "cpu_seconds": cb["seconds"], "cores": 1, "kind": "port",
                    "cpu_reference_shaped_seconds": cb.get("variants", {}).get("reference_shaped", {}).get("seconds"),
                    "gpu_ms_per_solve": ls["ms_per_solve"], "gpu_edges_per_s": ls["merged_edges"] * args.maxiter / (ls["ms_per_solve"] * 1e-3),
                    "gpu_over_cpu": cb["seconds"] / (ls["ms_per_solve"] * 1e-3),
                    "sample": "the WHOLE large_shop graph on both sides: 340 cameras x 10000 timesteps x 4 cameras per timestep, %d merged edges, "
                              "maxiter=%d + CG; CPU = the oracle's full solve on one host core (vectorised variant)" % (ls["merged_edges"], args.maxiter)}
        except Exception:
            pass
    if rank == 0:
        # the JSON line is the LAST thing on stdout: what C libraries printed into their own stdio buffer (RCCL's version banner
        # at communicator creation) is pushed out first - and the line itself is flushed here: a library's exit handler must
        # not be able to lose it (round 6 lost it that way once)
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
