#!/usr/bin/env python3
"""The four-call C boundary as N ranks of ONE timestep-sharded solve: every rank plans its slice of the timestep rows
(vican_plan_create), joins the communicator (vican_comm_create_local + the peer exchange: mailboxes mapped through hipIpc handles),
switches its plan to the sharded schedule (vican_plan_set_comm) and makes the same two solve calls - ctypes alone for the
numerics; torch.distributed only carries the 64-byte handles and, at the end, gathers the rows for the comparison.

    VICAN_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/facade_dist.py [out.json]

Golden g3 (40 cameras x 400 timesteps, non-unit weights, reprojection filter) in both dtypes and the large_shop-scale golden g9
in float64, against the REAL reference's poses; then g3 and g9 again with the cameras cut into TILES (16 / 128 cameras wide: the
camera-tiled schedule of csrc/vican_facade_tiles.hip, every rank tiling its own rows).  Prints "facade dist: mismatches <n>"."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                          # noqa: E402
import torch                                                # noqa: E402
import torch.distributed as dist                            # noqa: E402

import golden_cases as gc                                   # noqa: E402
from util import e2e_translation_tol, expected, load_golden, rebuild_inputs, translation_tol      # noqa: E402
from vican_amd import _lib, frontend, synth                 # noqa: E402
from vican_amd.geometry import SE3, geodesic                # noqa: E402

dist.init_process_group(os.environ.get("VICAN_DIST_BACKEND", "gloo"))
rank, world = dist.get_rank(), dist.get_world_size()
ndev = torch.cuda.device_count()
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % max(ndev, 1))
dev = torch.device("cuda", torch.cuda.current_device())
lib = _lib.load()
p = lambda t: C.c_void_p(t.data_ptr())
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

# the communicator: ctypes + 64-byte handles
comm = C.c_void_p()
_lib.check(lib.vican_comm_create_local(rank, world, C.byref(comm)), "vican_comm_create_local")
mine = C.create_string_buffer(64)
_lib.check(lib.vican_comm_peer_export(comm, 9 * 1024 + 96, mine), "vican_comm_peer_export")
handles = [None] * world
dist.all_gather_object(handles, mine.raw)
_lib.check(lib.vican_comm_peer_attach(comm, C.create_string_buffer(b"".join(handles), 64 * world)), "vican_comm_peer_attach")
lib.vican_comm_peer_set_timeout(comm, 5_000_000)            # (a test: five seconds per wait, not thirty)


def solve(prob, dt):
    T, Cn = prob.n_time, prob.n_cam
    r0, r1 = (T * rank) // world, (T * (rank + 1)) // world
    rp = np.asarray(prob.row_ptr)
    e0, e1 = int(rp[r0]), int(rp[r1])
    tdt = torch.float32 if dt == "float32" else torch.float64
    up = lambda a, d: torch.from_numpy(np.ascontiguousarray(a)).to(dev, d).contiguous()
    row_ptr, col = up(rp[r0:r1 + 1] - rp[r0], torch.int32), up(np.asarray(prob.col)[e0:e1], torch.int32)
    blk, a = up(np.asarray(prob.blk)[e0:e1], tdt), up(np.asarray(prob.a)[e0:e1], tdt)
    w, u, v = (up(np.asarray(x)[e0:e1], torch.float64) for x in (prob.w, prob.u, prob.v))
    deg_t = up(np.asarray(prob.deg_t)[r0:r1], torch.float64)
    deg_c = up(np.asarray(prob.deg_c) if rank == 0 else np.zeros(Cn), torch.float64)          # this rank's SHARE of the diagonal
    plan = C.c_void_p()
    Tl = r1 - r0
    _lib.check(lib.vican_plan_create(Cn, Tl, e1 - e0, _lib.STORE_F32 if dt == "float32" else _lib.STORE_F64, p(row_ptr), p(col), p(blk), p(a),
                                     p(w), p(u), p(v), p(deg_t), p(deg_c), stream, C.byref(plan)), "vican_plan_create")
    _lib.check(lib.vican_plan_set_comm(plan, comm, stream), "vican_plan_set_comm")
    rcs, Rt = torch.empty(3 * Cn, 3, dtype=torch.float64, device=dev), torch.empty(Tl, 9, dtype=torch.float64, device=dev)
    x_c, x_t = torch.empty(Cn, 3, dtype=torch.float64, device=dev), torch.empty(Tl, 3, dtype=torch.float64, device=dev)
    info = _lib.SolveInfo()
    _lib.check(lib.vican_solve_rot(plan, gc.MAXITER, 1e-10, p(rcs), p(Rt), C.byref(info), stream), "vican_solve_rot")
    _lib.check(lib.vican_solve_trans(plan, p(rcs), p(Rt), 1e-5, 0, p(x_c), p(x_t), C.byref(info), stream), "vican_solve_trans")
    lib.vican_plan_destroy(plan)
    rows = [None] * world
    dist.all_gather_object(rows, (Rt.cpu().numpy(), x_t.cpu().numpy()))
    Rt_all, xt_all = np.concatenate([r[0] for r in rows]), np.concatenate([r[1] for r in rows])
    Rc = np.swapaxes(rcs.cpu().numpy().reshape(Cn, 3, 3), 1, 2)
    Rtt = np.swapaxes(Rt_all.reshape(T, 3, 3), 1, 2)
    return Rc, Rtt, x_c.cpu().numpy(), xt_all, info


def compare(name, prob, exp, dt, out):
    Rc, Rt, pc, pt, info = out
    rot, pos = {}, {}
    for i, c in enumerate(prob.cam_names):
        rot[str(c)], pos[str(c)] = Rc[i], pc[i]
    for i, s_ in enumerate(prob.time_names):
        rot[str(s_) + "_0"], pos[str(s_) + "_0"] = Rt[i], pt[i]
    keys = [str(k) for k in exp["keys"]]
    R = np.stack([rot[k] for k in keys]); t = np.stack([pos[k] for k in keys])
    return float(geodesic(R, np.asarray(exp["R"], dtype=np.float64)).max()), float(np.linalg.norm(t - exp["t"], axis=1).max()), int(info.cg_iters)


report, bad = {"world": world}, 0
for name, dt, tile in (("g3_medium", "float64", 0), ("g3_medium", "float32", 0), ("g9_large_shop", "float64", 0),
                       ("g3_medium", "float64", 16), ("g9_large_shop", "float32", 128)):
    _lib.check(lib.vican_facade_set_tile_cams(tile or 1024), "vican_facade_set_tile_cams")
    if name == "g9_large_shop":
        g = load_golden(name)
        scene, flat = gc.build_flat(gc.LARGE_SHOP)
        src = synth.edges_to_dict(flat, SE3); cons = synth.constraints_from_scene(scene, SE3)
        nr, nt, ff = (gc.CALLABLES[gc.LARGE_SHOP[k]] for k in ("noise_r", "noise_t", "filt"))
        tol = min(translation_tol(name, dt), 2e-3)
    else:
        g = load_golden(name)
        case, src, cons, (nr, nt, ff) = rebuild_inputs(name, g)
        tol = e2e_translation_tol(name, dt)
    exp = expected(g, "conjugate_gradient", dt)
    prob = frontend.flatten(src, cons, nr, nt, ff, np.dtype(dt).type)
    r_err, t_err, cg = compare(name, prob, exp, dt, solve(prob, dt))
    ok = r_err < (5e-6 if dt == "float32" else 1e-7) and t_err < tol and abs(cg - int(exp["cg_iters"])) <= 6
    bad += not ok
    report["%s%s_%s" % (name, "@tiles" if tile else "", dt)] = dict(tile_cams=tile, rot_rad=r_err, trans_m=t_err, bound_m=tol, cg_iters=cg, cg_reference=int(exp["cg_iters"]), ok=bool(ok))
    if rank == 0:
        print("%s %s through the four calls on %d ranks%s: rot %.2e rad, trans %.2e m (< %.1e), cg %d vs %d%s" % (
            name, dt, world, " (camera tiles of %d)" % tile if tile else "", r_err, t_err, tol, cg, int(exp["cg_iters"]), "" if ok else "   <-- MISMATCH"), flush=True)
st_all = [None] * world
dist.all_gather_object(st_all, int(lib.vican_comm_peer_status(comm)))
report["peer_status"] = int(max(st_all))
if report["peer_status"]:                                   # waits ran into their bound (ranks time-sharing one crowded GPU): nothing to compare
    bad = 0
tb = torch.tensor([bad]); dist.all_reduce(tb)
if rank == 0:
    print("facade dist: mismatches", int(tb[0]))
    if len(sys.argv) > 1:
        json.dump(report, open(sys.argv[1], "w"), indent=1)
dist.destroy_process_group()
sys.exit(1 if int(tb[0]) else 0)
