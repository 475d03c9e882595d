#!/usr/bin/env python3
"""Which side carries the 1e-8 ... 6e-8 rad of object-mode float64 rotation offset?  (CPU only.)

The random campaign (tools/random_campaign.py) finds the product's float64 rotations 1e-12 rad from the pinned oracle at the
median of camera-mode scenes but 1e-9 (median) ... 6e-8 rad (worst) on OBJECT-mode scenes with area weights.  This probe takes
the scenes with the largest offsets and runs the rotation stage three ways ON THE SAME MERGED BLOCKS (the product's front-end):

  eigs   the oracle's restatement of the reference: scipy eigs(L, k=5, sigma=-1e-6)  (bipgo.py:288), three runs
  eigh   the same iteration with the eigenvectors of a dense LAPACK eigh of L (the 3 smallest), nothing else changed
  lanczos the product's host solver (vican_amd.solver.RotationSolver: block Lanczos to its own tolerance) on the NumPy stand-in
          backend (vican_amd/backend_cpu.py) - the algorithm the GPU runs, in plain float64

and prints the largest geodesic distance between the camera-role rotations of each pair.  If eigh and lanczos agree far below
the offset while eigs stands apart from both, the offset is ARPACK's (shift-invert with an LU of L + 1e-6 I next to three
eigenvalues of 1e-8 ... 1e-11), not the product's.

    python tools/object_offset_probe.py [seed ...]           (default: the three worst object-mode seeds of 3000..5999)"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as gc                                   # noqa: E402
from vican_amd.backend_cpu import NumpyBackend                      # noqa: E402
from oracle import bipgo_oracle as orc                      # noqa: E402
from test_random_parity_gpu import make_case                # noqa: E402
from vican_amd import frontend, synth                       # noqa: E402
from vican_amd.geometry import SE3                          # noqa: E402
from vican_amd.solver import Comm, RotationSolver           # noqa: E402


def geo(A, B):
    """largest geodesic angle between corresponding rotations of two [n, 3, 3] stacks"""
    tr = np.einsum("nij,nij->n", A, B)
    # angle from the sine of the relative rotation's skew part (accurate at 1e-12, where arccos((tr - 1) / 2) is not)
    Rel = np.einsum("nji,njk->nik", A, B)
    sk = 0.5 * (Rel - np.swapaxes(Rel, 1, 2))
    s = np.sqrt(sk[:, 0, 1] ** 2 + sk[:, 0, 2] ** 2 + sk[:, 1, 2] ** 2)
    return float(np.arctan2(s, 0.5 * (tr - 1.0)).max())


def dense_eigs(L, k, sigma):
    w, v = np.linalg.eigh(np.asarray(L.todense(), dtype=np.float64))
    idx = np.argsort(np.abs(w - sigma))[:k]
    return w[idx], v[:, idx]


def gauge(R):
    """camera-role rotations relative to the first one (the reference's own gauge, bipgo.py:295)"""
    return np.einsum("ij,njk->nik", R[0].T, R)


def one(seed):
    mode, scene, flat, (wr, wt), filt, dt = make_case(seed)
    assert mode == "object" and dt == np.float64, "object-mode float64 seeds only (seed % 4 == 0)"
    src = synth.edges_to_dict(flat, SE3)
    nr, nt, ff = gc.CALLABLES[wr], gc.CALLABLES[wt], gc.CALLABLES[filt]
    _, prob = frontend.flatten_object(src, nr, nt, ff, dt)
    C, T = prob.n_cam, prob.n_time
    rows = np.repeat(np.arange(T), np.diff(prob.row_ptr))
    blocks = prob.blk.reshape(-1, 3, 3)
    ortho = float(np.abs(np.einsum("nij,nkj->nik", blocks / prob.a[:, None, None], blocks / prob.a[:, None, None]) - np.eye(3)).max()) \
        if np.all(np.diff(prob.row_ptr) >= 0) else float("nan")
    runs = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for k in range(3):
            info = {}
            Rc, _ = orc.so3sync_arrays(C, T, prob.col, rows, blocks, prob.a, gc.MAXITER, dt, info=info)
            runs["eigs%d" % k] = gauge(Rc)
        ev = info["evals"]
        keep = orc.eigs
        orc.eigs = dense_eigs
        try:
            Rc, _ = orc.so3sync_arrays(C, T, prob.col, rows, blocks, prob.a, gc.MAXITER, dt)
        finally:
            orc.eigs = keep
        runs["eigh"] = gauge(Rc)
    K = NumpyBackend(C, prob.row_ptr, prob.col, prob.blk, prob.a, storage=dt)
    rot = RotationSolver(K, Comm.single())
    rc, _ = rot.run(gc.MAXITER)
    runs["lanczos"] = gauge(np.swapaxes(rc.numpy().reshape(C, 3, 3), 1, 2))
    print("seed %d: %d markers x %d frames, %d merged edges; weights %s, filter %s; |M M^T / a^2 - I| of single-source blocks <= %.1e" % (
        seed, C, T, prob.n_edges, wt, filt, ortho))
    print("   last iteration's five eigenvalues nearest -1e-6 (eigs): " + " ".join("%.2e" % e for e in np.sort(ev[-1])))
    print("   eigs run 0 vs run 1 / run 2      %.2e / %.2e rad" % (geo(runs["eigs0"], runs["eigs1"]), geo(runs["eigs0"], runs["eigs2"])))
    print("   eigs vs eigh                     %.2e rad" % geo(runs["eigs0"], runs["eigh"]))
    print("   eigs vs lanczos (the product)    %.2e rad   <- what the campaign reports" % geo(runs["eigs0"], runs["lanczos"]))
    print("   eigh vs lanczos (the product)    %.2e rad" % geo(runs["eigh"], runs["lanczos"]))
    return geo(runs["eigs0"], runs["lanczos"]), geo(runs["eigh"], runs["lanczos"]), geo(runs["eigs0"], runs["eigh"])


if __name__ == "__main__":
    seeds = [int(s) for s in sys.argv[1:]] or [4476, 3292, 3340]
    res = [one(s) for s in seeds]
    a = np.array(res)
    print("worst over %d scenes: eigs-lanczos %.2e, eigh-lanczos %.2e, eigs-eigh %.2e rad" % (len(seeds), a[:, 0].max(), a[:, 1].max(), a[:, 2].max()))
