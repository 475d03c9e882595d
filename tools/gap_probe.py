#!/usr/bin/env python3
"""Probe the idle gap around back-to-back sweep launches (run under rocprofv3 --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from vican_amd import synth
from vican_amd.device import HipBackend, LocalGraph
dev = torch.device("cuda:0")
gr = synth.make_merged_graph_torch(1000, 100000, 250, dev, torch.float32, seed=0)
g = LocalGraph(1000, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"])
K = HipBackend(g)
lamT, cd = K.empty(100000, 9), K.empty(1000)
K.init_duals(lamT, cd)
x = torch.linalg.qr(torch.randn(3000, 3, dtype=torch.float64, device=dev))[0].contiguous()
z = K.zeros(3000, 3)
Rt, lam2 = K.zeros(100000, 9), lamT.clone()
for rep in range(3):
    K.block_op_raw(lamT, x); K.block_op_raw(lamT, x); K.block_op_raw(lamT, x)       # MODE 0 x3
    K.fold_z(z); K.fold_z(z)
    K.dual_update(x, Rt, lam2); K.dual_update(x, Rt, lam2)                           # MODE 1 x2 (+svd, fx_finish)
    K.gauge_project(x, z); K.block_op_raw(lamT, x); K.gauge_project(x, z)
torch.cuda.synchronize()
