#!/usr/bin/env python3
"""Right-hand side J^T b (vican_trans_rhs: kernel + slab fold) launch time on a synthetic graph.
    python tools/rhs_time.py [--cams C --timesteps T --cpt K] [--layout wave|block] [--reps N]     (VICAN_WRHS_WAVES=4|8|12 for A/B)"""
import argparse
import sys
import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from vican_amd import synth                                     # noqa: E402
from vican_amd.device import HipBackend, LocalGraph             # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cams", type=int, default=1000)
ap.add_argument("--timesteps", type=int, default=100000)
ap.add_argument("--cpt", type=int, default=250)
ap.add_argument("--layout", default=None)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--tag", default="")
args = ap.parse_args()
C, T, K = args.cams, args.timesteps, args.cpt
dev = torch.device("cuda:0")
d = synth.make_merged_graph_torch(C, T, K, dev, torch.float32, seed=0)
g = LocalGraph(C, d["row_ptr"], d["col"], d["blk"], d["a"], d["w"], d["u"], d["v"], layout=args.layout)
H = HipBackend(g)
eye = torch.eye(3, dtype=torch.float64, device=dev)
rc, rt = eye.repeat(C, 1, 1).reshape(C, 9).contiguous(), eye.repeat(T, 1, 1).reshape(T, 9).contiguous()
b_t, b_c = H.empty(T, 3), H.empty(C, 3)
ts = []
for i in range(args.reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    H.trans_rhs(rc, rt, b_t, b_c)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
ts = np.array(ts[5:])
E = int(d["col"].numel())
by = 52 * E + 72 * T + 72 * C
print("%s layout %s n_wg %d: trans_rhs + fold: median %.1f us  min %.1f us  (%.0f MB algorithmic -> %.2f of 8 TB/s at the median)" % (
    args.tag, g.layout, g.n_wg, np.median(ts), ts.min(), by / 1e6, by / (np.median(ts) * 1e-6) / 8e12))
