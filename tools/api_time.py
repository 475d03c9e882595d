#!/usr/bin/env python3
"""Wall clock of the DROP-IN calls (edge dict in, pose dict out - what a user of the reference sees) on the reference's own
dataset shapes (cube_calib = object mode, 24 markers x <= 2000 frames; small_room = camera mode, <= 5000
timesteps; large_shop = 10 000 timesteps x hundreds of cameras), synthetic data of those shapes.  Per shape: first call, best of
the next three, the phases of that call, and - with --oracle - the CPU restatement of the reference on the same dict (once).
    python tools/api_time.py [--oracle]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from vican_amd import synth                                     # noqa: E402
from vican_amd.bipgo import bipartite_se3sync, object_bipartite_se3sync   # noqa: E402
from vican_amd.geometry import SE3                              # noqa: E402

from vican_amd.frontend import vectorized                      # noqa: E402

oracle = "--oracle" in sys.argv
unit, keep = (lambda e: 1.0), (lambda e: True)
# the same callables with a column form (called once for all edges instead of once per edge: frontend.vectorized)
vunit, vkeep = vectorized(lambda cols: 1.0)(lambda e: 1.0), vectorized(lambda cols: True)(lambda e: True)
SHAPES = [("cube_calib (object mode)", "object", 1, 2000, 24, 4), ("small_room", "camera", 12, 2000, 6, 3), ("small_room x 5000", "camera", 12, 5000, 6, 3),
          ("large_shop", "camera", 340, 10000, 6, 4)]
for name, mode, C, T, M, k in SHAPES:
    scene = synth.make_scene(n_cam=C, n_time=T, n_marker=M, seed=3)
    if mode == "camera":
        flat = synth.make_camera_edges(scene, cpt=k, mpv=2, sigma_r=1e-3, sigma_t=1e-3, seed=4)
        cons = synth.constraints_from_scene(scene, SE3)
        call = lambda s, inf: bipartite_se3sync(s, cons, unit, unit, keep, 4, "conjugate_gradient", np.float32, info=inf)
    else:
        flat = synth.make_object_edges(scene, mpv=k, sigma_r=1e-3, sigma_t=1e-3, seed=4)
        call = lambda s, inf: object_bipartite_se3sync(s, unit, unit, keep, 4, "conjugate_gradient", np.float32, info=inf)
    src = synth.edges_to_dict(flat, SE3)
    times = []
    for i in range(4):
        info = {}
        t0 = time.perf_counter()
        out = call(src, info)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0, info))
    best, binfo = min(times[1:], key=lambda x: x[0])
    line = "%-26s %6d source edges, %4d x %5d nodes, %6d merged edges: first call %7.1f ms, then %6.1f ms (front-end %.1f, pack %.2f, rotations %.2f, translations %.2f; cg %s)" % (
        name, len(src), binfo["n_cam"], binfo["n_time"], binfo["n_edges"], times[0][0] * 1e3, best * 1e3,
        (best - binfo["t_pack"] - binfo["t_rot"] - binfo["t_trans"]) * 1e3, binfo["t_pack"] * 1e3, binfo["t_rot"] * 1e3, binfo["t_trans"] * 1e3, binfo["cg_iters"])
    if mode == "camera":
        vt = []
        for i in range(4):
            t0 = time.perf_counter()
            bipartite_se3sync(src, cons, vunit, vunit, vkeep, 4, "conjugate_gradient", np.float32)
            torch.cuda.synchronize()
            vt.append(time.perf_counter() - t0)
        line += "; callables with a column form: %.1f ms" % (min(vt[1:]) * 1e3)
    if oracle:
        from oracle import bipgo_oracle as orc
        t0 = time.perf_counter()
        if mode == "camera":
            orc.bipartite_se3sync(src, cons, unit, unit, keep, 4, "conjugate_gradient", np.float32, loop=True)
        else:
            orc.object_bipartite_se3sync(src, unit, unit, keep, 4, "conjugate_gradient", np.float32, loop=True)
        line += "; reference restated on one CPU core %.1f s" % (time.perf_counter() - t0)
    print(line, flush=True)
