#!/usr/bin/env python3
"""Five cold solve_problem calls on the large_shop shape (for a rocprofv3 --kernel-trace + tools/timeline.py)."""
import sys
import numpy as np
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from vican_amd import bipgo, frontend, synth                    # noqa: E402
from vican_amd.geometry import SE3                              # noqa: E402
scene = synth.make_scene(n_cam=340, n_time=10000, n_marker=6, seed=0)
flat = synth.make_camera_edges(scene, cpt=4, mpv=1, sigma_r=1e-3, sigma_t=1e-3, seed=1)
cams = flat["cam_key"].astype(str)
tm = np.char.partition(flat["marker_key"].astype(str), "_")
cons = synth.constraints_from_scene(scene, SE3)
ones = np.ones(len(cams))
prob = frontend.flatten_arrays(cams, tm[:, 0], tm[:, 2], flat["R"], flat["t"], ones, ones, cons, np.float32)
import time
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    info = {}
    t0 = time.perf_counter()
    bipgo.solve_problem(prob, 4, "conjugate_gradient", np.float32, info=None)
    torch.cuda.synchronize()
    print("call %d: %.2f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
