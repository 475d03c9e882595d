#!/usr/bin/env python3
"""Operator-sweep launch time on a RAGGED sparse capture: C cameras, T timesteps, every timestep seen by a uniformly random
number of cameras in [lo, hi] (real captures are ragged; `bench.py --workload sparse` has exactly 8 per timestep), for both
orders of the edges inside a chunk (VICAN_SLOT_ORDER).   python tools/ragged_time.py [--cams 100 --timesteps 2000000 --lo 2 --hi 8]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from vican_amd import synth                                     # noqa: E402
from vican_amd.device import HipBackend, LocalGraph             # noqa: E402



ap = argparse.ArgumentParser()
ap.add_argument("--cams", type=int, default=100)
ap.add_argument("--timesteps", type=int, default=2000000)
ap.add_argument("--lo", type=int, default=2)
ap.add_argument("--hi", type=int, default=8)
ap.add_argument("--reps", type=int, default=40)
args = ap.parse_args()
C, T, lo, hi = args.cams, args.timesteps, args.lo, args.hi
dev = torch.device("cuda:0")
row_ptr, col, blk, a32 = synth.make_ragged_graph_torch(C, T, lo, hi, dev)
E = int(col.numel())
xq = torch.linalg.qr(torch.randn(3 * C, 3, dtype=torch.float64, device=dev))[0].contiguous()
for so in ("banks", "rows", None):
    if so is None:
        os.environ.pop("VICAN_SLOT_ORDER", None)
    else:
        os.environ["VICAN_SLOT_ORDER"] = so
    gr = LocalGraph(C, row_ptr, col, blk, a32)
    H = HipBackend(gr)
    lam, dg, zz = H.empty(T, 9), H.empty(C), H.empty(3 * C, 3)
    H.init_duals(lam, dg)
    ts = []
    for i in range(args.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); H.block_op_raw(lam, xq); e1.record()
        H.fold_z(zz)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = np.array(ts[5:])
    print("C=%d T=%d rows of %d..%d edges (E=%d) order %-7s layout %s slots/edges %.3f max_rows %d: median %.1f us (%.0f GB/s = %.3f of 8 TB/s), %.2f ps/edge" % (
        C, T, lo, hi, E, so or "default", gr.layout, gr.padded_slots() / E, gr.max_rows, np.median(ts), gr.op_bytes() / np.median(ts) * 1e-3,
        gr.op_bytes() / np.median(ts) * 1e-3 / 8000, np.median(ts) * 1e6 / E))
    del H, gr
