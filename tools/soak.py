#!/usr/bin/env python3
"""Soak: thousands of WHOLE warm solves (rotation loop + translation CG, Python driver) per shape, every output compared with the
first solve's bit for bit on the device, Lanczos step counts and CG iterations with it, cooperative-kernel failures listed.

    python tools/soak.py [scale]  (GPU; ~1 minute at scale 1)

Output of the round-6 library: profiles/r06_determinism.txt (second half)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vican_amd import synth
from vican_amd.device import make_backend
from vican_amd.solver import Comm, RotationSolver, TranslationSolver
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
SCALE = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0        # python tools/soak.py 15  -> fifteen times as many solves
for (C, T, cpt, n_sol, tag) in ((340, 10000, 4, 20000, "large_shop"), (1000, 100000, 250, 4000, "stress"), (100, 200000, 8, 4000, "sparse-like"), (4000, 30000, 250, 1500, "tiled")):
    n_sol = int(n_sol * SCALE)
    gr = synth.make_merged_graph_torch(C, T, cpt, dev, torch.float32, seed=0)
    g, K = make_backend(C, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"])
    del gr
    rot, tr = RotationSolver(K, Comm()), TranslationSolver(K, Comm())
    def solve():
        rot.stats = dict(sweeps=0, lanczos_steps=[], evals=[], restarts=0, resid=[])
        rc, Rt = rot.run(4)
        tr.setup(rc, Rt)
        xc, xt = tr.solve(3 * (C + T))
        return rc, Rt, xc, xt
    for _ in range(10):                 # (the check positions of capture-sized graphs settle over the first solves: solver.py, spectral)
        out = solve()
    K.synchronize()
    ref = [o.clone() for o in out]
    steps0, it0 = list(rot.stats["lanczos_steps"]), tr.info.get("cg_iters")
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    odd = 0
    t0 = time.perf_counter()
    for i in range(n_sol):
        out = solve()
        m = torch.zeros((), dtype=torch.bool, device=dev)
        for o, r in zip(out, ref):
            m = m | (o != r).any()
        bad += m
        if rot.stats["lanczos_steps"] != steps0 or tr.info.get("cg_iters") != it0:
            odd += 1
            if odd <= 5:
                print("  %s solve %d: lanczos %s cg %s (usual %s / %s)" % (tag, i, rot.stats["lanczos_steps"], tr.info.get("cg_iters"), steps0, it0), flush=True)
    K.synchronize()
    print("%s: %d solves, %d with other step counts, %d whose outputs differ from the first by a bit; %.2f ms per solve (with the comparison); failures of cooperative kernels: %s"
          % (tag, n_sol, odd, int(bad), (time.perf_counter() - t0) / n_sol * 1e3, getattr(K, "coop_failures", [])), flush=True)
    del K, g, rot, tr
    torch.cuda.empty_cache()
