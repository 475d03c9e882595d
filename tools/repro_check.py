#!/usr/bin/env python3
"""Repeat one operator sweep on the stress graph and report bitwise reproducibility (GPU box)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from vican_amd import synth
from vican_amd.device import HipBackend, LocalGraph
dev = torch.device("cuda:0")
C, T = 1000, 100000
gr = synth.make_merged_graph_torch(C, T, 250, dev, torch.float32, seed=9, sigma_r=0.0, sigma_t=0.0)
g = LocalGraph(C, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"])
K = HipBackend(g)
lamT, cd = K.empty(T, 9), K.empty(C)
K.init_duals(lamT, cd)
x = torch.linalg.qr(torch.randn(3 * C, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1)))[0].contiguous()
z0 = K.zeros(3 * C, 3)
K.block_op(lamT, x, z0)
bad = 0
zprev = None
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    z = K.zeros(3 * C, 3)
    K.block_op(lamT, x, z)
    if zprev is not None and not torch.equal(z, zprev):
        print("run %d differs from run %d" % (i, i - 1))
    zprev = z
    if not torch.equal(z, z0):
        d = (z - z0).abs()
        bad += 1
        cols = (d.reshape(C, 9) > 0).sum(0).tolist()
        print("   differing entries per component q:", cols)
        print("run %d differs: max abs %.3e (rel %.3e), %d entries, sched %s" % (i, float(d.max()), float(d.max() / z0.abs().max()), int((d > 0).sum()),
              g.fx.view(torch.int32)[20:22].tolist()))
print("mismatching runs vs the first:", bad)
