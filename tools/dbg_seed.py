import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as gc
from oracle import bipgo_oracle as orc
from test_random_parity_gpu import make_case
from vican.bipgo import bipartite_se3sync, object_bipartite_se3sync
from vican_amd import synth
from vican_amd.geometry import SE3, geodesic
seed = int(sys.argv[1])
mode, scene, flat, (wr, wt), filt, dt = make_case(seed)
src = synth.edges_to_dict(flat, SE3)
nr, nt, ff = gc.CALLABLES[wr], gc.CALLABLES[wt], gc.CALLABLES[filt]
info, oinfo = {}, {}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    if mode == "camera":
        cons = synth.constraints_from_scene(scene, SE3)
        res = bipartite_se3sync(src, cons, nr, nt, ff, gc.MAXITER, "conjugate_gradient", dt, info=info)
        ref = orc.bipartite_se3sync(src, cons, nr, nt, ff, gc.MAXITER, "conjugate_gradient", dt, loop=True, info=oinfo)
    else:
        res = object_bipartite_se3sync(src, nr, nt, ff, gc.MAXITER, "conjugate_gradient", dt, info=info)
        ref = orc.object_bipartite_se3sync(src, nr, nt, ff, gc.MAXITER, "conjugate_gradient", dt, loop=True, info=oinfo)
R = np.stack([np.asarray(res[k].R(), dtype=np.float64) for k in res]); Rr = np.stack([np.asarray(ref[k].R(), dtype=np.float64) for k in ref])
print(os.environ.get("VICAN_LANCZOS_RESIDENT", "1"), os.environ.get("VICAN_CG_RESIDENT", "1"), "rot %.3e" % float(geodesic(R, Rr).max()),
      "steps", info.get("lanczos_steps"), "resid", ["%.1e" % r for r in info.get("eig_resid", [])], "early", info.get("early_exit"), "layout", info.get("layout"))
