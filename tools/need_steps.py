import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vican_amd import synth
from vican_amd.device import HipBackend, LocalGraph
from vican_amd.solver import Comm, RotationSolver
dev = torch.device("cuda:0")
for C, T, k in ((340, 10000, 4), (40, 400, 3), (1000, 100000, 250)):
    gr = synth.make_merged_graph_torch(C, T, k, dev, torch.float32, seed=0)
    g = LocalGraph(C, gr["row_ptr"], gr["col"], gr["blk"], gr["a"])
    K = HipBackend(g)
    rot = RotationSolver(K, Comm.single())
    rot.min_steps, rot.warm_min_steps, rot.check_every, rot.small_graph = 1, 1, 1, True
    rot.run(4)
    print("C=%d T=%d k=%d: needed steps (checked every step) %s  resid %s" % (C, T, k, rot.stats["lanczos_steps"], ["%.1e" % r for r in rot.stats["resid"]]))
    rot2 = RotationSolver(HipBackend(g), Comm.single()); rot2.run(4)
    print("    default schedule: %s resid %s" % (rot2.stats["lanczos_steps"], ["%.1e" % r for r in rot2.stats["resid"]]))
