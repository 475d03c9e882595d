#!/usr/bin/env python3
"""Per-phase wall clock of the resident CG kernel (VICAN_CGRSTAMP build: VICAN_LIB=.../libvican_hip_cgrst.so)."""
import sys
import numpy as np
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from vican_amd import synth
from vican_amd.device import HipBackend, LocalGraph
from vican_amd.solver import TranslationSolver
C, T, K = 340, 10000, 4
dev = torch.device("cuda:0")
d = synth.make_merged_graph_torch(C, T, K, dev, torch.float32, seed=0)
g = LocalGraph(C, d["row_ptr"], d["col"], d["blk"], d["a"], d["w"], d["u"], d["v"])
H = HipBackend(g)
S = TranslationSolver(H, rtol=1e-5)
eye = torch.eye(3, dtype=torch.float64, device=dev)
S.setup(eye.repeat(C, 1, 1).reshape(C, 9).contiguous(), eye.repeat(T, 1, 1).reshape(T, 9).contiguous())
for _ in range(3):
    S.solve(3 * (C + T))
torch.cuda.synchronize()
ws = H._cgres_ws.cpu().numpy()
o = ws[9 * C + 4 * H.cgl.n_wg + 1: 9 * C + 4 * H.cgl.n_wg + 11] / 100.0
names = ["beta,p update", "sweep", "slab store + pq", "barrier 1", "fold slice", "barrier 2", "gather + q_c + p.q", "step + reductions", "barrier 3", "gather rr"]
print("iterations", S.info["cg_iters"], "n_wg", H.cgl.n_wg)
for n, v in zip(names, o):
    print("  %-22s %6.2f us / iteration" % (n, v))
print("  sum %.2f" % o.sum())
