#!/usr/bin/env python3
"""Per-phase ticks (10 ns) of the cooperative camera-side Lanczos step on a large_shop-sized graph (COOP_STAMP build:
VICAN_LIB=.../libvican_hip_coopst.so; the kernel prints from workgroup 3)."""
import sys
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
V, HB, xrow, beta0 = H.zeros(3 * (m + 1) * n), H.zeros(m, hw + 9), H.empty(n, 3), H.empty(9)
R, Hs, G, z = H.zeros(3 * n), H.zeros(3 * (m + 1) * 3), H.zeros(9), H.empty(n, 3)
H.lanczos_seed(x0, V, ld, beta0, xrow)
for j in range(m):
    H.block_op_slabs(lam, xrow)
    H.lanczos_cam_step(lamC, V, ld, j, z, R, Hs, G, HB[j, :hw], HB[j, hw:], xrow, 0.0, from_slabs=True)
torch.cuda.synchronize()
print("phases: stage+fold+AQ | gram1 | barrier | reduce+update -> gram2 | barrier | reduce+update+G | barrier | chol+write")
