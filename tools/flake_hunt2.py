#!/usr/bin/env python3
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests"); sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np, torch
import test_kernels_gpu as t
C, T, lo, hi, bt, nwg, er = t.CONFIGS[5]
for trial in range(6):
    H, N, g = t.make_backends(C, T, lo, hi, 100 + C, np.float32, bt, nwg, er)
    rng = np.random.default_rng(1)
    x = np.linalg.qr(rng.standard_normal((3 * C, 3)))[0]
    lam = rng.standard_normal((T, 3, 3)); lam = lam @ np.swapaxes(lam, 1, 2) + np.eye(3)
    lam_h = H.from_numpy(lam.reshape(T, 9)); xh = H.from_numpy(x)
    H.set_duals(lam_h)
    outs = []
    zp = []
    for i in range(40):
        z = H.empty(3 * C, 3)
        H.block_op(lam_h, xh, z)
        torch.cuda.synchronize()
        outs.append(z.cpu().numpy().copy()); zp.append(H.zpart.view(torch.int64)[: g.n_wg * 9 * C].cpu().numpy().copy())
    ref = outs[0]
    fxv = g.fx.cpu().numpy()
    nd = [int((o != ref).sum()) for o in outs]
    md = [float(np.abs(o - ref).max()) for o in outs]
    tot = [p.reshape(g.n_wg, -1).sum(0) for p in zp]
    ndt = [int((q != tot[0]).sum()) for q in tot]
    print("trial", trial, "n_chunk", g.n_chunk, "cap", g.wg_chunk_cap, "max_rows", g.max_rows, "z res", fxv[3] * fxv[7], "differing entries", nd, "max diff %.3e" % max(md), "| slab totals differing:", ndt)
    if max(nd):
        k = int(np.argmax(nd)); d = (outs[k] - ref)
        idx = np.argwhere(d != 0)[:6]
        print("   sample diffs:", [(tuple(i), float(d[tuple(i)]), float(d[tuple(i)] / (fxv[3] * fxv[7]))) for i in idx])
