#!/usr/bin/env python3
"""Per-phase, per-step wall clock of the resident Lanczos kernel (VICAN_LRSTAMP build: VICAN_LIB=.../libvican_hip_lrst.so)."""
import sys
import numpy as np
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from vican_amd import synth
from vican_amd.device import HipBackend, LocalGraph
C, T, K, m = 340, 10000, 4, 12
dev = torch.device("cuda:0")
d = synth.make_merged_graph_torch(C, T, K, dev, torch.float32, seed=0)
g = LocalGraph(C, d["row_ptr"], d["col"], d["blk"], d["a"], d["w"], d["u"], d["v"])
H = HipBackend(g)
n, ld, hw = 3 * C, 3 * C, 3 * (m + 1) * 3
lam, cd, lamC = H.empty(T, 9), H.empty(C), H.empty(C, 9)
H.init_duals(lam, cd); H.scaled_identity(cd, lamC)
x0 = torch.randn(n, 3, dtype=torch.float64, device=dev)
for rep in range(3):
    V, HB, xrow, beta0 = H.zeros(3 * (m + 1) * n), H.zeros(m, hw + 9), H.empty(n, 3), H.empty(9)
    H.lanczos_seed(x0, V, ld, beta0, xrow)
    H.lanczos_resident(lam, lamC, V, ld, 0, m, xrow, HB, hw, 0.0)
torch.cuda.synchronize()
ws = H._lres_ws.cpu().numpy()
ncw = -(-C // 32)
base = ncw * (2 * 3 * 192 + 8) + 1
o = ws[base: base + 12 * m].reshape(m, 12) / 100.0
names = ["stage x", "sweep", "slab store", "barrier 1", "fold+AQ+gram", "barrier 2", "reduce+upd+gram", "barrier 3", "reduce+upd+G", "barrier 4", "chol+write", "barrier 5"]
print("n_wg", g.n_wg, "ncw", ncw)
print("%-18s" % "phase" + "".join("%7d" % j for j in range(m)))
for i, nme in enumerate(names):
    print("%-18s" % nme + "".join("%7.2f" % o[j, i] for j in range(m)))
print("%-18s" % "sum" + "".join("%7.2f" % o[j].sum() for j in range(m)))
