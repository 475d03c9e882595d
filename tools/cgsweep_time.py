#!/usr/bin/env python3
"""CG Laplacian product (vican_cg_sweep) launch time on a synthetic graph, optionally through a diagnostic build of
the library (VICAN_LIB=<variant .so>, see tools/build_variants.py).

    python tools/cgsweep_time.py [--cams C --timesteps T --cpt K] [--layout wave|block] [--reps N]"""
import argparse
import sys
import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from vican_amd import synth                                     # noqa: E402
from vican_amd.device import HipBackend, LocalGraph                      # noqa: E402
from vican_amd.solver import TranslationSolver                  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cams", type=int, default=1000)
ap.add_argument("--timesteps", type=int, default=100000)
ap.add_argument("--cpt", type=int, default=250)
ap.add_argument("--layout", default=None)
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--tag", default="")
ap.add_argument("--stamp", action="store_true")
args = ap.parse_args()
C, T, K = args.cams, args.timesteps, args.cpt
dev = torch.device("cuda:0")
d = synth.make_merged_graph_torch(C, T, K, dev, torch.float32, seed=0)
g = LocalGraph(C, d["row_ptr"], d["col"], d["blk"], d["a"], d["w"], d["u"], d["v"], layout=args.layout)
H = HipBackend(g)
S = TranslationSolver(H, rtol=1e-30)
eye = torch.eye(3, dtype=torch.float64, device=dev)
rc, rt = eye.repeat(C, 1, 1).reshape(C, 9).contiguous(), eye.repeat(T, 1, 1).reshape(T, 9).contiguous()
S.setup(rc, rt)
S.solve(3 * (C + T), maxiter=3)                                  # leaves a live CG state (not done, not first)
ts, tk = [], []
for i in range(args.reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ka, kb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for e in (ka, kb):
        e.record()                                                # (creates the HIP event; rebound to the kernel's dispatch below)
    H.cg_begin(S.r_c, S.p_c, 1e-30, S.st)
    a.record()
    H.time_next_sweep((ka, kb))                                   # HIP events bound to the sweep kernel's own dispatch
    H.cg_sweep(S.deg_t, S.p_c, S.r_t, S.p_t, S.q_t, S.qcpq, S.st)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
    tk.append(ka.elapsed_time(kb) * 1e3)
ts, tk = np.array(ts[5:]), np.array(tk[5:])
E = int(d["col"].numel())
by = 12 * E + 4 * (T + 1) + 96 * T + 48 * C                       # SURVEY 8(d) / DESIGN section 5: 12E + 4(T+1) + 96T + 48C
print("%s layout %s n_wg %d: cg sweep kernel alone: median %.1f us  min %.1f us = %.2f of 8 TB/s | + slab fold + pq reduce: median %.1f us  (%.0f MB algorithmic)" % (
    args.tag, H.cgl.kind, H.cgl.n_wg, np.median(tk), tk.min(), by / np.median(tk) / 8e6, np.median(ts), by / 1e6))

if args.stamp:
    import ctypes
    from vican_amd import _lib
    nw, nwg = g.wg_waves, H.cgl.n_wg
    buf = torch.zeros(nwg * nw * 10, dtype=torch.float64, device=dev)
    lib = _lib.load()
    lib.vican_cgw_stamp_buffer.restype, lib.vican_cgw_stamp_buffer.argtypes = ctypes.c_int, [ctypes.c_void_p]
    _lib.check(lib.vican_cgw_stamp_buffer(buf.data_ptr()), "stamp buffer")
    for _ in range(3):
        H.cg_begin(S.r_c, S.p_c, 1e-30, S.st)
        H.cg_sweep(S.deg_t, S.p_c, S.r_t, S.p_t, S.q_t, S.qcpq, S.st)
    torch.cuda.synchronize()
    full = buf.cpu().numpy().reshape(nwg, nw, 10)
    r = full[:, :, :4] / 100.0
    r -= r[:, :, 0].min()
    print("    wave start %.1f .. %.1f | loop start %.1f .. %.1f (median %.1f) | wave loop end: min %.1f median %.1f max %.1f | kernel end %.1f" % (
        r[:, :, 0].min(), r[:, :, 0].max(), r[:, :, 1].min(), r[:, :, 1].max(), np.median(r[:, :, 1]), r[:, :, 2].min(), np.median(r[:, :, 2]),
        r[:, :, 2].max(), r[:, :, 3].max()))
    ph = full[:, :, 4:9].sum((0, 1)) / full[:, :, 9].sum()
    print("    per chunk and wave (s_memtime ticks): issue %.0f | wait for chunk data %.0f | commit %.0f | edges %.0f | fold %.0f | sum %.0f ; chunks per wave %.1f" % (
        ph[0], ph[1], ph[2], ph[3], ph[4], ph.sum(), full[:, :, 9].mean()))
    wg_end = r[:, :, 2].max(1)
    print("    last-wave loop end by blockIdx mod 8: " + " ".join("%.1f" % wg_end[i::8].mean() for i in range(8)))
