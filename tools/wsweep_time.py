#!/usr/bin/env python3
"""Operator sweep (vican_block_op) launch time on a synthetic graph, per layout / launch shape, optionally through a
diagnostic build of the library (VICAN_LIB=<variant .so>, see tools/build_variants.py).

    python tools/wsweep_time.py [--cams C --timesteps T --cpt K] [--dtype f32|f64] variant ...
    variant = wave[:wg_waves[:n_copy[:n_wg]]] | block[:block_threads[:n_copy[:n_wg]]]      ('-' = default)
With a VICAN_WSTAMP build (--stamp) the wall-clock structure of the launch is printed."""
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
ap.add_argument("--dtype", default="f32")
ap.add_argument("--stamp", action="store_true")
ap.add_argument("--ragged", default=None, help="lo:hi - every timestep is seen by a uniformly random number of cameras in [lo, hi] (f32)")
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("variants", nargs="*", default=["wave", "block"])
args = ap.parse_args()
C, T, K = args.cams, args.timesteps, args.cpt
dev = torch.device("cuda:0")
tdt = torch.float32 if args.dtype == "f32" else torch.float64
if args.ragged:
    lo_, hi_ = (int(v) for v in args.ragged.split(":"))
    rp_, col_, blk_, a_ = synth.make_ragged_graph_torch(C, T, lo_, hi_, dev)
    d = dict(row_ptr=rp_, col=col_, blk=blk_.to(tdt), a=a_.to(tdt))
else:
    d = synth.make_merged_graph_torch(C, T, K, dev, tdt, seed=0)
x = torch.linalg.qr(torch.randn(3 * C, 3, dtype=torch.float64, device=dev))[0].contiguous()
for var in args.variants:
    f = var.split(":")
    p = [None if (i >= len(f) or f[i] in ("", "-")) else int(f[i]) for i in range(1, 4)]
    try:
        if f[0] == "wave":
            g = LocalGraph(C, d["row_ptr"], d["col"], d["blk"], d["a"], layout="wave", wg_waves=p[0], n_copy=p[1], n_wg=p[2])
        else:
            g = LocalGraph(C, d["row_ptr"], d["col"], d["blk"], d["a"], layout="block", block_threads=p[0], n_copy=p[1], n_wg=p[2])
    except Exception as e:                                      # noqa: BLE001
        print(var, "->", e); continue
    H = HipBackend(g)
    lam, deg, z = H.empty(T, 9), H.empty(C), H.empty(3 * C, 3)
    H.init_duals(lam, deg)
    ts = []
    for i in range(args.reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); H.block_op_raw(lam, x); b.record()
        H.fold_z(z)                                              # a small kernel in between, as in the solver
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts = np.array(ts[5:])
    print("%-18s layout %-5s threads %4d n_copy %2d n_wg %3d max_rows %3d chunks %6d pad %.3f: median %.1f us  min %.1f us  (%.0f GB/s = %.3f of 8 TB/s at median)" % (
        var, g.layout, g.block_threads, g.n_copy, g.n_wg, g.max_rows, g.n_chunk, g.padded_slots() / max(g.n_edges, 1), np.median(ts), ts.min(),
        g.op_bytes() / np.median(ts) * 1e-3, g.op_bytes() / np.median(ts) * 1e-3 / 8000))
    if args.stamp and g.layout == "wave":
        import ctypes
        from vican_amd import _lib
        from vican_amd.device import _ptr, _stream
        nw = g.wg_waves
        buf = torch.zeros(g.n_wg * nw * 10, dtype=torch.float64, device=dev)
        lib = _lib.load()
        fn = lib.vican_block_op_stamp
        fn.restype, fn.argtypes = ctypes.c_int, [ctypes.POINTER(_lib.Graph)] + [ctypes.c_void_p] * 6
        for _ in range(3):
            _lib.check(fn(ctypes.byref(g.desc), _ptr(lam), _ptr(x), _ptr(H.zpart), _ptr(g.fx), _ptr(buf), _stream()), "stamp")
        torch.cuda.synchronize()
        full = buf.cpu().numpy().reshape(g.n_wg, nw, 10)
        r = full[:, :, :4] / 100.0     # us
        t0 = r[:, :, 0].min()
        r -= t0
        print("    first/last wave start %.1f / %.1f | loop start %.1f .. %.1f | wave loop end: min %.1f median %.1f max %.1f | "
              "per-WG last wave end: min %.1f median %.1f max %.1f | kernel end %.1f" % (
                  r[:, :, 0].min(), r[:, :, 0].max(), r[:, :, 1].min(), r[:, :, 1].max(), r[:, :, 2].min(), np.median(r[:, :, 2]),
                  r[:, :, 2].max(), r[:, :, 2].max(1).min(), np.median(r[:, :, 2].max(1)), r[:, :, 2].max(1).max(), r[:, :, 3].max()))
        ph = full[:, :, 4:9].sum((0, 1)) / full[:, :, 9].sum() / nw * nw          # cycles per chunk per wave (s_memtime ticks)
        nchunk_w = g.n_chunk / (g.n_wg * nw)
        ph = full[:, :, 4:9].mean((0, 1)) / nchunk_w
        print("    per chunk and wave (s_memtime ticks): issue %.0f | wait for chunk data %.0f | phase 1 %.0f | phase 2 %.0f | phase 3 + LDS drain %.0f | sum %.0f" % (
            ph[0], ph[1], ph[2], ph[3], ph[4], ph.sum()))
        done = full[:, :, 9].sum(1)
        print("    chunks processed per workgroup: min %d median %d max %d (total %d of %d) | by blockIdx mod 8: %s" % (
            done.min(), np.median(done), done.max(), done.sum(), g.n_chunk, " ".join("%.0f" % done[i::8].mean() for i in range(8))))
        wg_end = r[:, :, 2].max(1)
        print("    last-wave loop end by blockIdx mod 8: " + " ".join("%.1f" % wg_end[i::8].mean() for i in range(8)) +
              " | by blockIdx / 32: " + " ".join("%.1f" % wg_end[32 * i:32 * i + 32].mean() for i in range(g.n_wg // 32)))
        print("    spread of wave loop ends inside a WG: median %.1f us, max %.1f us" % (
            np.median(r[:, :, 2].max(1) - r[:, :, 2].min(1)), (r[:, :, 2].max(1) - r[:, :, 2].min(1)).max()))
    del H, g
