#!/usr/bin/env python3
"""Operator sweep (vican_block_op) on the stress graph for different launch shapes:
python tools/sweep_variants.py  -> median / min launch time per (block_threads, n_copy, n_wg)."""
import sys
import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from vican_amd import synth                                     # noqa: E402
from vican_amd.device import HipBackend, LocalGraph             # noqa: E402

C, T, K = 1000, 100000, 250
dev = torch.device("cuda:0")
d = synth.make_merged_graph_torch(C, T, K, dev, torch.float32, seed=0)
x = torch.linalg.qr(torch.randn(3 * C, 3, dtype=torch.float64, device=dev))[0].contiguous()
variants = [(768, None, None), (512, None, None), (1024, None, None), (768, 4, None), (768, 16, None), (512, None, 512), (768, None, 512)]
if len(sys.argv) > 1:
    variants = [tuple(None if v == "-" else int(v) for v in a.split(",")) for a in sys.argv[1:]]
for bt, ncopy, nwg in variants:
    try:
        g = LocalGraph(C, d["row_ptr"], d["col"], d["blk"], d["a"], block_threads=bt, n_copy=ncopy, n_wg=nwg)
    except Exception as e:                                      # noqa: BLE001
        print(bt, ncopy, nwg, "->", e); continue
    H = HipBackend(g)
    lam, deg, z = H.empty(T, 9), H.empty(C), H.empty(3 * C, 3)
    H.init_duals(lam, deg)
    ts = []
    for i in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); H.block_op_raw(lam, x); b.record()
        H.fold_z(z)                                              # a small kernel in between, as in the solver
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts = np.array(ts[5:])
    print("block %4d n_copy %2d n_wg %3d max_rows %2d chunks %5d: median %.1f us  min %.1f us  (%.0f GB/s at median)" % (
        g.block_threads, g.n_copy, g.n_wg, g.max_rows, g.n_chunk, np.median(ts), ts.min(), g.op_bytes() / np.median(ts) * 1e-3))
    del H, g
