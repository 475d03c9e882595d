#!/usr/bin/env python3
"""Time the translation stage with lsqr_solver="direct" (GPU LSQR) vs CG on synthetic graphs (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from vican_amd import synth
from vican_amd.device import HipBackend, LocalGraph
from vican_amd.solver import Comm, RotationSolver, TranslationSolver, LsqrTranslationSolver
dev = torch.device("cuda:0")
for (C, T, k) in ((340, 10000, 4), (1000, 100000, 250)):
    gr = synth.make_merged_graph_torch(C, T, k, dev, torch.float32, seed=0)
    g = LocalGraph(C, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"], keep_csr=True)
    K = HipBackend(g)
    rot = RotationSolver(K, Comm()); rc, Rt = rot.run(4)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr = TranslationSolver(K, Comm()); tr.setup(rc, Rt); tr.solve(3 * (C + T))
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ls = LsqrTranslationSolver(K, Comm()); ls.solve(rc, Rt, 3 * (C + T), None)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        K.lsqr_host_scalars = True                  # the round-2 path: Golub-Kahan scalars on the host, two edge passes per iteration
        lh = LsqrTranslationSolver(K, Comm()); lh.solve(rc, Rt, 3 * (C + T), None)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        K.lsqr_host_scalars = False
        # exactly 40 iterations, no stopping test can fire: the cost of one iteration without set-up / polling / cancelled launches
        lf = LsqrTranslationSolver(K, Comm(), atol=0.0, btol=0.0, conlim=0.0)
        lf.solve(rc, Rt, 3 * (C + T), None, iter_lim=8)
        torch.cuda.synchronize(); t4 = time.perf_counter()
        lf.solve(rc, Rt, 3 * (C + T), None, iter_lim=8)
        torch.cuda.synchronize(); t5 = time.perf_counter()
        n8 = lf.info.get("lsqr_iters")
        lf.solve(rc, Rt, 3 * (C + T), None, iter_lim=40)
        torch.cuda.synchronize(); t6 = time.perf_counter()
        n40 = lf.info.get("lsqr_iters")      # (may end early: LSQR converges to machine precision - istop 5 - in ~19 iterations on the stress graph)
        print("    no stopping tolerance: %d iterations %.2f ms, %d iterations %.2f ms (istop %s) -> %.3f ms per iteration" % (
            n8, (t5 - t4) * 1e3, n40, (t6 - t5) * 1e3, lf.info.get("istop"), (t6 - t5 - (t5 - t4)) * 1e3 / max(n40 - n8, 1)))
        print("C=%d T=%d E=%d: CG %.2f ms (%d it)   LSQR device scalars + fused pass %.2f ms (%s it, istop %s: %.3f ms / it)   host scalars %.2f ms (%s it: %.3f ms / it)" % (
            C, T, g.n_edges, (t1 - t0) * 1e3, tr.info["cg_iters"], (t2 - t1) * 1e3, ls.info.get("lsqr_iters"), ls.info.get("istop"),
            (t2 - t1) * 1e3 / max(ls.info.get("lsqr_iters"), 1), (t3 - t2) * 1e3, lh.info.get("lsqr_iters"), (t3 - t2) * 1e3 / max(lh.info.get("lsqr_iters"), 1)))
