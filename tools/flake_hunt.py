#!/usr/bin/env python3
"""Hunt for run-to-run nondeterminism: every edge-side entry point, on every test graph shape and both storage
types, launched `reps` times on identical inputs - outputs must be bit-identical (exact fixed-point sums, fixed
reduction orders).  python tools/flake_hunt.py [reps]"""
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests"); sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
import torch
import test_kernels_gpu as t
from vican_amd import synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
extra = [(1000, 4000, 200, 300, None, None, False), (340, 10000, 2, 6, None, None, False), (600, 3000, 20, 60, None, None, False)]
bad = 0
for cfg in t.CONFIGS + extra:
    C, T, lo, hi, bt, nwg, er = cfg
    for dt in (np.float32, np.float64):
        H, N, g = t.make_backends(C, T, lo, hi, 100 + C, dt, bt, nwg, er)
        rng = np.random.default_rng(1)
        x = H.from_numpy(np.linalg.qr(rng.standard_normal((3 * C, 3)))[0])
        xn = H.from_numpy(np.linalg.qr(rng.standard_normal((3 * (C + T), 3)))[0])
        rc = H.from_numpy(synth.random_rotations(rng, C).reshape(3 * C, 3))
        rt = H.from_numpy(synth.random_rotations(rng, max(T, 1)).reshape(-1, 9))
        lam0, cd = H.empty(T, 9), H.empty(C)
        H.init_duals(lam0, cd)

        def run_all():
            out = {}
            z = H.empty(3 * C, 3); H.block_op(lam0, x, z); out["block_op"] = z
            Rt, L, zr = H.zeros(T, 9), H.zeros(T, 9), H.empty(3 * C, 3)
            H.dual_update_op(rc, Rt, L, zr); out["dual_update_op.z"] = zr; out["dual_update_op.Rt"] = Rt; out["dual_update_op.L"] = L
            Rt2, L2 = H.zeros(T, 9), H.zeros(T, 9)
            H.dual_update(rc, Rt2, L2); out["dual_update.Rt"] = Rt2
            H.init_duals(lam0, cd)                                  # restore the scales for block_op
            zb = H.empty(3 * (C + T), 3); H.bip_scales(); H.bip_apply(xn, zb); out["bip_apply"] = zb
            H.init_duals(lam0, cd)
            rhs_t, rhs_c = H.zeros(max(T, 1), 3), H.zeros(C, 3)
            H.trans_rhs(rc, rt, rhs_t, rhs_c); out["trans_rhs.t"] = rhs_t; out["trans_rhs.c"] = rhs_c
            torch.cuda.synchronize()
            return {k: torch.nan_to_num(v.clone(), nan=-7.0, posinf=-8.0, neginf=-9.0) for k, v in out.items()}

        ref = run_all()
        for i in range(reps):
            cur = run_all()
            for k in ref:
                if not torch.equal(ref[k], cur[k]):
                    bad += 1
                    d = (ref[k] - cur[k]).abs()
                    print("DIFF cfg %s %s %s rep %d: %d entries, max %.3e" % (cfg, dt.__name__, k, i, int((d > 0).sum()), float(d.max())))
        del H, g
print("done: %d differing outputs over %d repetitions per case" % (bad, reps))
