#!/usr/bin/env python3
"""Camera-side Lanczos step: cooperative multi-workgroup kernel vs single-workgroup fused kernel, per camera count
(full rotation stage of a large_shop-like graph, solver object reused)."""
import sys, time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch
from vican_amd import synth
from vican_amd.device import HipBackend, LocalGraph
from vican_amd.solver import Comm, RotationSolver
dev = torch.device("cuda:0")
for C, T, k in ((100, 5000, 4), (340, 10000, 4), (600, 10000, 6), (1000, 20000, 8)):
    gr = synth.make_merged_graph_torch(C, T, k, dev, torch.float32, seed=0)
    g = LocalGraph(C, gr["row_ptr"], gr["col"], gr["blk"], gr["a"])
    for coop in (True, False):
        K = HipBackend(g); K.coop_cam_step = coop
        rot = RotationSolver(K, Comm.single())
        ts = []
        for rep in range(8):
            rot.stats = dict(sweeps=0, lanczos_steps=[], evals=[], restarts=0, resid=[])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rot.run(4)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print("C=%4d T=%5d: %-11s rotation stage %.3f ms (min of 8), lanczos steps %s" % (
            C, T, "cooperative" if coop else "single-WG", 1e3 * min(ts[2:]), rot.stats["lanczos_steps"]))
