"""One campaign seed repeated: is a disagreement the product's or the oracle's (ARPACK's start vector is random)?"""
import sys
sys.path.insert(0, __file__.rsplit("/", 3)[0]); sys.path.insert(0, __file__.rsplit("/", 3)[0] + "/tests")
import numpy as np
import golden_cases as gc
from test_random_parity_gpu import make_case
from vican_amd import synth
from vican_amd.bipgo import bipartite_se3sync, object_bipartite_se3sync
from vican_amd.geometry import SE3, geodesic
from oracle import bipgo_oracle as orc
seed = int(sys.argv[1])
mode, scene, flat, (wr, wt), filt, dt = make_case(seed)
src = synth.edges_to_dict(flat, SE3)
nr, nt, ff = gc.CALLABLES[wr], gc.CALLABLES[wt], gc.CALLABLES[filt]
cons = synth.constraints_from_scene(scene, SE3) if mode == "camera" else None
def product():
    if mode == "camera":
        return bipartite_se3sync(src, cons, nr, nt, ff, gc.MAXITER, "conjugate_gradient", dt)
    return object_bipartite_se3sync(src, nr, nt, ff, gc.MAXITER, "conjugate_gradient", dt)
def oracle():
    info = {}
    if mode == "camera":
        r = orc.bipartite_se3sync(src, cons, nr, nt, ff, gc.MAXITER, "conjugate_gradient", dt, loop=True, info=info)
    else:
        r = orc.object_bipartite_se3sync(src, nr, nt, ff, gc.MAXITER, "conjugate_gradient", dt, loop=True, info=info)
    return r, info
R = lambda d: np.stack([np.asarray(d[k].R(), dtype=np.float64) for k in d])
p0 = product()
prods = [R(product()) for _ in range(3)]
print("product repeats bit-identical:", all(np.array_equal(R(p0), x) for x in prods))
ors = []
for i in range(6):
    try:
        o, info = oracle()
        ors.append(R(o))
        print("oracle run %d: vs product %.3e rad, vs oracle run 0 %.3e rad, first evals %s" % (i, geodesic(R(p0), ors[-1]).max(), geodesic(ors[0], ors[-1]).max(), np.array(info["evals"])[0]))
    except Exception as e:                                       # noqa: BLE001
        print("oracle run %d raised %r" % (i, e))
