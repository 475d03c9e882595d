#!/usr/bin/env python3
"""Time vican_ritz alone for a few basis sizes (GPU box)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from test_kernels_gpu import make_backends, _ritz_inputs

H, N, g = make_backends(5, 40, 1, 3, 7, np.float64)
m = 32
for steps in (2, 3, 4, 5, 6, 8, 12, 16, 24, 26, 32):
    HB, hw = _ritz_inputs(steps, m, 50 + steps)
    HBd = H.from_numpy(HB)
    Yd, std, gd = H.zeros(3 * (m + 1), 3), H.zeros(16), H.zeros(1, dtype=torch.int32)
    for _ in range(3):
        H.ritz(HBd, hw, steps, 1, 1e-10, 1e-7, -1.0, Yd, std, gd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        H.ritz(HBd, hw, steps, 1, 1e-10, 1e-7, -1.0, Yd, std, gd)
    e1.record(); torch.cuda.synchronize()
    extra = ""
    if "--stamp" in sys.argv and steps < m:
        st = Yd.cpu().numpy().reshape(-1)[9 * steps: 9 * steps + 5] / 100.0
        extra = "   stamps (us): tridiag %.1f gersh %.1f bisect %.1f invit %.1f backtr+valid %.1f" % tuple(st)
    print("steps %2d n %3d: %.1f us per call%s" % (steps, 3 * steps, e0.elapsed_time(e1) / 20 * 1e3, extra))
