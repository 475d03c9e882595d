#!/usr/bin/env python3
"""Where the host front-end of a large_shop-sized drop-in call spends its time (80 000 source edges, dict of SE3 objects):
the phases of frontend.flatten timed one by one, with per-edge (scalar) and column-form callables.
    python tools/frontend_phases.py"""
import itertools
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from vican_amd import frontend, synth                           # noqa: E402
from vican_amd.geometry import SE3                              # noqa: E402

scene = synth.make_scene(n_cam=340, n_time=10000, n_marker=6, seed=3)
flat = synth.make_camera_edges(scene, cpt=4, mpv=2, sigma_r=1e-3, sigma_t=1e-3, seed=4)
src = synth.edges_to_dict(flat, SE3)
cons = synth.constraints_from_scene(scene, SE3)
unit, keep = (lambda e: 1.0), (lambda e: True)


def best(f, k=5):
    out = 1e9
    for _ in range(k):
        t0 = time.perf_counter(); r = f(); out = min(out, time.perf_counter() - t0)
    return out * 1e3, r


n = len(src)
print("%d source edges" % n)
t, keys = best(lambda: list(src.keys())); t2, vals = best(lambda: list(src.values()))
print("keys + values lists          %6.2f ms" % (t + t2))
t, _ = best(lambda: np.array([bool(keep(v)) for v in vals], dtype=bool)); print("edge_filter per edge         %6.2f ms" % t)
t, _ = best(lambda: np.array([unit(v) for v in vals], dtype=np.float64)); print("one noise model per edge     %6.2f ms" % t)
cols = frontend.EdgeColumns(vals)
t, poses = best(lambda: [v["pose"] for v in vals]); print("pose objects                 %6.2f ms" % t)
t, Rl = best(lambda: [p.R() for p in poses]); print("pose.R() per edge            %6.2f ms" % t)
t, _ = best(lambda: frontend._stack_f64(Rl, (3, 3))); print("stack R [n,3,3]              %6.2f ms" % t)
t, tl = best(lambda: [p.t() for p in poses]); print("pose.t() per edge            %6.2f ms" % t)
t, _ = best(lambda: frontend._stack_f64(tl, (3,))); print("stack t [n,3]                %6.2f ms" % t)
t, cams = best(lambda: [k[0] for k in keys]); print("camera ids                   %6.2f ms" % t)
t, tm = best(lambda: np.array([k[1] for k in keys])); print("'<t>_<m>' strings -> array   %6.2f ms" % t)
t, parts = best(lambda: np.char.partition(tm, "_")); print("split at '_' (vectorised)    %6.2f ms" % t)
t, codes = best(lambda: frontend.index_codes(cams, parts[:, 0], parts[:, 2])); print("ids -> indices               %6.2f ms" % t)
t, _ = best(lambda: keys == list(keys)); print("same keys as last call?      %6.2f ms" % t)
t, ix = best(lambda: frontend.index_edges(None, None, None, cons, codes=codes)); print("constraint tables            %6.2f ms" % t)
nomerge = lambda ix, R, t, kr, kt, dt: None
vunit, vkeep = frontend.vectorized(lambda c: 1.0)(unit), frontend.vectorized(lambda c: True)(keep)
for name, fns in (("scalar callables", (unit, unit, keep)), ("column-form callables", (vunit, vunit, vkeep))):
    def go(reset):
        if reset:
            frontend._CODES_CACHE["keys"] = None
        frontend.flatten(src, cons, fns[0], fns[1], fns[2], np.float32, merge=nomerge)
    t_new, _ = best(lambda: go(True)); t_same, _ = best(lambda: go(False))
    print("flatten without the merge, %-22s new keys %6.2f ms, keys of the previous call %6.2f ms" % (name + ":", t_new, t_same))
