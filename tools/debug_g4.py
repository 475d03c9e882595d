import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from test_parity_gpu import *
import golden_cases as gc
for name in ("g4_illcond", "g3_medium", "g2_small"):
    for dt in ("float64", "float32"):
        from vican.bipgo import bipartite_se3sync, object_bipartite_se3sync
        g = load_golden(name)
        case, src, cons, (nr, nt, ff) = rebuild_inputs(name, g)
        exp = expected(g, "conjugate_gradient", dt)
        info = {}
        res = bipartite_se3sync(src, constraints=cons, noise_model_r=nr, noise_model_t=nt, edge_filter=ff,
                                maxiter=gc.MAXITER, lsqr_solver="conjugate_gradient", dtype=np.dtype(dt).type, info=info)
        rot, tr = pose_errors(res, exp)
        print(name, dt, "rot %.3e tr %.3e cg %d (ref %d) steps %s resid %s" % (rot, tr, info["cg_iters"], int(exp["cg_iters"]), info["lanczos_steps"], ["%.1e" % r for r in info["eig_resid"]]))
