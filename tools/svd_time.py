#!/usr/bin/env python3
"""Timing of the T-sized dual kernels (init_duals, dual_svd) alone and behind a cache-evicting kernel."""
import sys; sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch, numpy as np
from vican_amd import synth, _lib
from vican_amd.device import HipBackend, LocalGraph, _ptr, _stream
dev = torch.device("cuda:0")
d = synth.make_merged_graph_torch(1000, 100000, 250, dev, torch.float32, seed=0)
g = LocalGraph(1000, d["row_ptr"], d["col"], d["blk"], d["a"]); H = HipBackend(g)
T = 100000
lam, deg = H.empty(T, 9), H.empty(1000)
big = torch.randn(512 * 1024 * 1024 // 8, device=dev, dtype=torch.float64)

def t(fn, pre=None, n=20):
    ts = []
    for i in range(n):
        if pre: pre()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return np.median(ts[3:])

f_init = lambda: H.lib.vican_init_duals(T, _ptr(g.row_sum_a), _ptr(g.rnorm), _ptr(lam), _ptr(g.fx), _stream())
print("event pair alone %.1f us" % t(lambda: None))
print("init_duals warm %.1f us | after 512 MB eviction %.1f us" % (t(f_init), t(f_init, lambda: big.add_(1.0))))
rc = (d["R_cam"].transpose(1, 2).contiguous().reshape(3000, 3) if "R_cam" in d else torch.eye(3, dtype=torch.float64, device=dev).repeat(1000, 1)).contiguous()
Rt = H.empty(T, 9)
H.init_duals(lam, deg)
f_du = lambda: H.dual_update(rc, Rt, lam)
print("dual_update (sweep + dual_svd + fx_finish) %.1f us" % t(f_du))
