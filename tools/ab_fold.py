import sys, time
sys.path.insert(0, '/root/repo')
import torch, numpy as np
from vican_amd import synth
from vican_amd.device import HipBackend, LocalGraph
from vican_amd.solver import Comm, RotationSolver
dev = torch.device("cuda:0")
for C, T, k in ((340, 10000, 4), (1000, 100000, 250)):
    gr = synth.make_merged_graph_torch(C, T, k, dev, torch.float32, seed=0)
    g = LocalGraph(C, gr["row_ptr"], gr["col"], gr["blk"], gr["a"])
    res = {}
    for rnd in range(3):
        for fold in (True, False):
            K = HipBackend(g); rot = RotationSolver(K, Comm.single()); rot.fold_in_step = fold
            ts = []
            for rep in range(12):
                rot.stats = dict(sweeps=0, lanczos_steps=[], evals=[], restarts=0, resid=[])
                torch.cuda.synchronize(); t0 = time.perf_counter(); rot.run(4); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            res.setdefault(fold, []).append(1e3 * np.median(ts[3:]))
    print("C=%d T=%d: rotation stage, fold in step %s ms | separate fold kernel %s ms" % (C, T, ["%.3f" % v for v in res[True]], ["%.3f" % v for v in res[False]]))
