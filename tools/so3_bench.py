#!/usr/bin/env python3
"""Time the non-eliminated variant (bipartite_so3sync machinery) on the benchmark's stress graph:
the one-pass bipartite operator (sweep MODE 2) against the eliminated operator (MODE 0), and a whole
GeneralRotationSolver run.  python tools/so3_bench.py [n_cam n_time cams_per_t]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from vican_amd import synth                                     # noqa: E402
from vican_amd.device import HipBackend, LocalGraph             # noqa: E402
from vican_amd.solver import Comm, GeneralRotationSolver, RotationSolver   # noqa: E402

C, T, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1000, 100000, 250)
dev = torch.device("cuda:0")
d = synth.make_merged_graph_torch(C, T, K, dev, torch.float32, seed=0)
g = LocalGraph(C, d["row_ptr"], d["col"], d["blk"], d["a"], layout="block")
H = HipBackend(g)
N = C + T
x = torch.linalg.qr(torch.randn(3 * N, 3, dtype=torch.float64, device=dev))[0].contiguous()
z = H.empty(3 * N, 3)
lam = H.empty(T, 9); deg = H.empty(C)


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3


H.bip_scales()
t2 = timed(lambda: H.bip_apply(x, z))
H.init_duals(lam, deg)
xc = x[: 3 * C].contiguous(); zc = H.empty(3 * C, 3)
t0 = timed(lambda: H.block_op(lam, xc, zc))
byt = g.op_bytes()
print("graph C=%d T=%d E=%d | bip_apply (MODE 2 + slab fold) %.1f us | block_op (MODE 0 + slab fold) %.1f us | algorithmic bytes %d"
      % (C, T, g.n_edges, t2, t0, byt))
print("  MODE 2 moves the same blocks plus x_time/y_time (2 x T x 72 B): %.0f GB/s" % ((byt + T * 72) / t2 * 1e-3))

for name, cls in (("eliminated (large_bipartite_so3sync)", RotationSolver), ("non-eliminated (bipartite_so3sync)", GeneralRotationSolver)):
    rot = cls(H, Comm.single())
    for rep in range(3):
        rot.stats = dict(sweeps=0, lanczos_steps=[], evals=[], restarts=0, resid=[])
        torch.cuda.synchronize(); t = time.perf_counter()
        rot.run(4)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("%-40s maxiter=4: %.2f ms  sweeps %d  lanczos steps %s  resid %s" % (
        name, dt * 1e3, rot.stats["sweeps"], rot.stats["lanczos_steps"], ["%.1e" % r for r in rot.stats["resid"]]))
    print("   smallest eigenvalues per iteration:", np.array2string(np.array(rot.stats["evals"])[:, :3], precision=3))
