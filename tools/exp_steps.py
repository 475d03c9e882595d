import sys, os
sys.path.insert(0, "/root/repo")
os.chdir("/root/repo")
import torch, numpy as np, time
from vican_amd import synth
from vican_amd.device import HipBackend, LocalGraph
from vican_amd.solver import Comm, RotationSolver
dev = torch.device("cuda:0")
gr = synth.make_merged_graph_torch(1000, 100000, 250, dev, torch.float32, seed=0, t_offset=0)
g = LocalGraph(1000, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"])
K = HipBackend(g)
for ms, ce, wm in ((6, 2, 2), (4, 1, 2), (4, 1, 1)):
    rot = RotationSolver(K, Comm(), min_steps=ms, check_every=ce, warm_min_steps=wm)
    for rep in range(3):
        rot.stats = dict(sweeps=0, lanczos_steps=[], evals=[], restarts=0, resid=[])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc, Rt = rot.run(4)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(ms, ce, wm, "rep", rep, "steps", rot.stats["lanczos_steps"], "resid", ["%.1e" % r for r in rot.stats["resid"]], "%.2f ms" % (dt * 1e3), "checks", rot.stats.get("n_check"))
    print("rc checksum", float(rc.abs().sum()))
