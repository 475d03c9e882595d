import sys, torch, numpy as np
sys.path.insert(0, '.')
from vican_amd import synth, solver
from vican_amd.device import HipBackend, LocalGraph
from vican_amd.solver import Comm, RotationSolver
dev = torch.device('cuda:0')
gr = synth.make_merged_graph_torch(340, 10000, 4, dev, torch.float32, seed=0)
g = LocalGraph(340, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"])
print("bt", g.block_threads, "ncopy", g.n_copy, "max_rows", g.max_rows, "chunks", g.n_chunk)
K = HipBackend(g)
rot = RotationSolver(K, Comm())
orig = rot._project
def proj(steps):
    out = orig(steps)
    th, Y, res, scale, breakdown, eff = out
    print("  check steps=%d eff=%d r=%.3e th0..2=%s" % (steps, eff, res.max() / scale, th[:3]))
    return out
rot._project = proj
for rep in range(2):
    print("solve", rep)
    rot.stats = dict(sweeps=0, lanczos_steps=[], evals=[], restarts=0, resid=[])
    rot.run(4)
    print(rot.stats["lanczos_steps"], rot.stats["restarts"], rot.pred_steps)
