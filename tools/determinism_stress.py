#!/usr/bin/env python3
"""Run-to-run determinism of the hot path's kernels: every kernel's sums are fixed-point (exact integers) or fixed-order, so repeated
launches on the same inputs must reproduce the first result BIT FOR BIT - whatever order the workgroups ran in.  N repetitions per
kernel and shape, compared on the device; shapes: the stress benchmark, large_shop, a sparse capture, the camera-tiled path.

    python tools/determinism_stress.py [N=20000]          (GPU; ~1 minute at the default)

Round 6 found a scheduler race with this (wave_sweep_kernel: 2 of 20 000 operator sweeps, 11 of 10 000 fused dual updates missed a
chunk - DESIGN.md, "Fixed on the way"); the output of the fixed library is profiles/r06_determinism.txt."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vican_amd import synth
from vican_amd.device import HipBackend, LocalGraph, make_backend
from vican_amd.solver import Comm, RotationSolver, TranslationSolver
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
def stress(name, fn, outs, n=N, prep=None):
    if prep: prep()
    fn(); torch.cuda.synchronize()
    ref = [o.clone() for o in outs]
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    for rep in range(n):
        if prep: prep()
        fn()
        m = torch.zeros((), dtype=torch.bool, device=dev)
        for o, r in zip(outs, ref):
            m = m | ((o != r) & ~((o != o) & (r != r))).any()
        cnt += m
    torch.cuda.synchronize()
    print("%s: %d deviating repetitions of %d" % (name, int(cnt), n), flush=True)

for (C, T, cpt, tag) in ((1000, 100000, 250, "stress"), (340, 10000, 4, "large_shop"), (100, 200000, 8, "sparse-like")):
    gr = synth.make_merged_graph_torch(C, T, cpt, dev, torch.float32, seed=0)
    K = HipBackend(LocalGraph(C, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"]))
    del gr
    rot = RotationSolver(K, Comm())
    rc, Rt = rot.run(4); K.synchronize()
    rc, Rt = rc.clone(), Rt.clone()
    n = N if tag == "stress" else 3 * N
    # operator + dual update on this shape
    x = rot.xrow.clone(); z = K.empty(3 * C, 3)
    stress(tag + " block_op", lambda: K.block_op(rot.lamT, x, z), [z], n)
    Rt2, lamT2, zraw = K.empty(T, 9), K.empty(T, 9), K.empty(3 * C, 3)
    stress(tag + " dual_update_op", lambda: K.dual_update_op(rc, Rt2, lamT2, zraw), [Rt2, lamT2, zraw], n // 2)
    K.set_duals(rot.lamT)
    # right-hand side
    tr = TranslationSolver(K, Comm())
    bt, bc = K.empty(max(T, 1), 3), K.empty(C, 3)
    stress(tag + " trans_rhs", lambda: K.trans_rhs(rc, Rt, bt, bc), [bt, bc], n // 2)
    # whole CG solve (fused iterations / resident kernel): x_c, x_t and the iteration count
    tr.setup(rc, Rt)
    def solve():
        tr.solve(3 * (C + T))
    stress(tag + " CG solve", solve, [tr.x_c, tr.x_t, tr.st], max(200, n // 20))
    del K, rot, tr
    torch.cuda.empty_cache()

# the tiled operator (4 tiles) and its CG product
C, T, cpt = 4000, 30000, 250
gr = synth.make_merged_graph_torch(C, T, cpt, dev, torch.float32, seed=0)
g, K = make_backend(C, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"])
del gr
rot = RotationSolver(K, Comm())
rc, Rt = rot.run(4); K.synchronize()
x = rot.xrow.clone(); z = K.empty(3 * C, 3)
stress("tiled block_op (fused launch)", lambda: K.block_op(rot.lamT, x, z), [z], N)
Rt2, lamT2 = K.empty(T, 9), rot.lamT.clone()
stress("tiled dual_update", lambda: K.dual_update(rc, Rt2, lamT2), [Rt2, lamT2], N // 4)
K.set_duals(rot.lamT)
tr = TranslationSolver(K, Comm())
tr.setup(rc.clone(), Rt.clone())
stress("tiled CG solve", lambda: tr.solve(3 * (C + T)), [tr.x_c, tr.x_t, tr.st], max(100, N // 40))
