"""Randomised end-to-end parity on the GPU: drop-in API vs the oracle (pinned against the real reference by
tests/test_oracle_golden.py) on seeded random scenes of varying shape, noise, weights, filter and dtype - the
combinations the fixed goldens do not enumerate."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import golden_cases as gc                                   # noqa: E402
from vican_amd import synth                                 # noqa: E402
from vican_amd.bipgo import DisconnectedGraphWarning      # noqa: E402
from vican_amd.geometry import SE3, geodesic                # noqa: E402


def make_case(seed):
    rng = np.random.default_rng(1000 + seed)
    mode = "camera" if seed % 4 else "object"
    n_cam = 1 if mode == "object" else int(rng.integers(2, 30))
    n_time = int(rng.integers(20, 400))
    n_marker = int(rng.integers(2, 12)) if mode == "camera" else int(rng.integers(4, 24))
    scene = synth.make_scene(n_cam=n_cam, n_time=n_time, n_marker=n_marker, seed=seed)
    sig = float(10.0 ** rng.uniform(-4, -2))
    if mode == "camera":
        flat = synth.make_camera_edges(scene, cpt=int(min(n_cam, rng.integers(2, 5))), mpv=int(rng.integers(1, 4)),
                                       sigma_r=sig, sigma_t=sig, seed=seed + 1)
    else:
        flat = synth.make_object_edges(scene, mpv=int(rng.integers(2, 6)), sigma_r=sig, sigma_t=sig, seed=seed + 1)
    # dtype by seed % 2, weights by the next bit (the bit after that for object scenes, which are the seeds = 0 mod 4): all
    # four combinations occur - rounds 1-2 tied f32 to the area weights and f64 to unit weights
    weights = [("w_unit", "w_unit"), ("w_area_mild", "w_area_mild_t")][(seed // 4 if mode == "object" else seed // 2) % 2]
    filt = "f_err" if seed % 3 == 0 else "f_all"
    dt = np.float32 if seed % 2 else np.float64
    return mode, scene, flat, weights, filt, dt


@pytest.mark.parametrize("seed", [5, 10, 15, 20, 25, 30])
def test_random_scene_direct_solver_matches_oracle(seed):
    """lsqr_solver="direct" (scipy LSQR semantics on the GPU) on the same random scenes."""
    from oracle import bipgo_oracle as orc
    from vican.bipgo import bipartite_se3sync, object_bipartite_se3sync
    mode, scene, flat, (wr, wt), filt, dt = make_case(seed)
    src = synth.edges_to_dict(flat, SE3)
    nr, nt, ff = gc.CALLABLES[wr], gc.CALLABLES[wt], gc.CALLABLES[filt]
    info = {}
    if mode == "camera":
        cons = synth.constraints_from_scene(scene, SE3)
        res = bipartite_se3sync(src, cons, nr, nt, ff, gc.MAXITER, "direct", dt, info=info)
        try:
            ref = orc.bipartite_se3sync(src, cons, nr, nt, ff, gc.MAXITER, "direct", dt, loop=True)
        except TypeError:
            assert info["n_cam"] <= 2
            return
    else:
        res = object_bipartite_se3sync(src, nr, nt, ff, gc.MAXITER, "direct", dt, info=info)
        ref = orc.object_bipartite_se3sync(src, nr, nt, ff, gc.MAXITER, "direct", dt, loop=True)
    t = np.stack([np.asarray(res[k].t(), dtype=np.float64) for k in res])
    tr = np.stack([np.asarray(ref[k].t(), dtype=np.float64) for k in ref])
    err = float(np.linalg.norm(t - tr, axis=1).max())
    # LSQR stops on atol = btol = 1e-6 (relative): the iterates of two implementations agree far below that accuracy
    assert err < (1e-5 if dt == np.float64 else 1e-3) * (1.0 + float(np.abs(tr).max())), (seed, err, info.get("lsqr_iters"))


@pytest.mark.parametrize("seed", range(12))
def test_random_scene_matches_oracle(seed):
    from oracle import bipgo_oracle as orc
    from vican.bipgo import bipartite_se3sync, object_bipartite_se3sync
    mode, scene, flat, (wr, wt), filt, dt = make_case(seed)
    src = synth.edges_to_dict(flat, SE3)
    nr, nt, ff = gc.CALLABLES[wr], gc.CALLABLES[wt], gc.CALLABLES[filt]
    info, oinfo = {}, {}
    from util import SelfMovement

    def rotations_agree(ref):
        Rp = np.stack([np.asarray(res[k].R(), dtype=np.float64) for k in res])
        Rq = np.stack([np.asarray(ref[k].R(), dtype=np.float64) for k in ref])
        return float(geodesic(Rp, Rq).max()) < (1e-7 if dt == np.float64 else 5e-6)
    with SelfMovement(orc) as sm:       # the oracle's cg call + 8 repeats on right-hand sides perturbed by one unit in the last place
        if mode == "camera":
            cons = synth.constraints_from_scene(scene, SE3)
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                res = bipartite_se3sync(src, cons, nr, nt, ff, gc.MAXITER, "conjugate_gradient", dt, info=info)
            if any(issubclass(w.category, DisconnectedGraphWarning) for w in caught):
                # the reprojection filter cut the graph into pieces (seed 729: 54 edges, 4 components): the reference
                # leaves its loop after one iteration (max_eval <= 1e-6, bipgo.py:283) with an arbitrary null-space
                # basis; nothing to compare - the drop-in has warned and still returns finite poses
                assert all(np.isfinite(p.R()).all() and np.isfinite(p.t()).all() for p in res.values())
                return
            try:
                ref = orc.bipartite_se3sync(src, cons, nr, nt, ff, gc.MAXITER, "conjugate_gradient", dt, loop=True, info=oinfo)
            except TypeError:
                # one or two cameras: the reference's eigs(k=5) needs k < 3C - 1 and raises (scipy: "Cannot use
                # scipy.linalg.eig for sparse A with k >= N - 1"); the product solves these (deliberately more capable)
                assert info["n_cam"] <= 2
                assert all(np.isfinite(p.R()).all() and np.isfinite(p.t()).all() for p in res.values())
                return
        else:
            res = object_bipartite_se3sync(src, nr, nt, ff, gc.MAXITER, "conjugate_gradient", dt, info=info)
            ref = orc.object_bipartite_se3sync(src, nr, nt, ff, gc.MAXITER, "conjugate_gradient", dt, loop=True, info=oinfo)
    assert [str(k) for k in res] == [str(k) for k in ref]
    if not rotations_agree(ref):
        # the reference's eigs call starts ARPACK from a random vector and goes astray about once in a thousand runs (rotations
        # 1e-4 rad off, fine on the next run of the same input: tools/dbg/seed_repeat.py): one more oracle run
        with SelfMovement(orc) as sm:
            if mode == "camera":
                ref = orc.bipartite_se3sync(src, cons, nr, nt, ff, gc.MAXITER, "conjugate_gradient", dt, loop=True, info=oinfo)
            else:
                ref = orc.object_bipartite_se3sync(src, nr, nt, ff, gc.MAXITER, "conjugate_gradient", dt, loop=True, info=oinfo)
    R = np.stack([np.asarray(res[k].R(), dtype=np.float64) for k in res])
    Rr = np.stack([np.asarray(ref[k].R(), dtype=np.float64) for k in ref])
    t = np.stack([np.asarray(res[k].t(), dtype=np.float64) for k in res])
    tr = np.stack([np.asarray(ref[k].t(), dtype=np.float64) for k in ref])
    rot = float(geodesic(R, Rr).max())
    assert rot < (1e-7 if dt == np.float64 else 5e-6), (seed, rot)
    # translations: bounded by the oracle's OWN reproducibility - 4 x the largest distance its answer moves when its
    # right-hand side is perturbed (8 trials), floored at 1e-6 m.  (Rounds 1-2 bounded this by 5 x the oracle's distance to
    # the CONVERGED solution of its system, which is metres wide on weighted scenes.)  Perturbation size: end to end the two
    # translation stages are fed rotations that differ by `rot` (two eigen-solvers: ~1e-9 rad in f64, 1e-7 .. 1e-6 in f32), which
    # moves the right-hand side by that much relative - so the trials perturb by max(rot, 1e-15); tools/random_campaign.py also
    # runs the stage alone on the oracle's rotations against the 1e-15 yardstick, and the NumPy stand-in for calibration.
    err = float(np.linalg.norm(t - tr, axis=1).max())
    tol = max(sm.bound(), 4.0 * float(sm.more_trials(max(rot, 1e-15)).max()))
    assert err < tol, (seed, err, rot, sm.self_move.tolist(), info["cg_iters"], sm.iters.tolist())
    lo, hi = int(sm.iters.min()), int(sm.iters.max())
    assert lo - 2 <= info["cg_iters"] <= hi + 2, (seed, info["cg_iters"], sm.iters.tolist())


@pytest.mark.parametrize("maxiter", [1, 2, 3, 6])
@pytest.mark.parametrize("seed", [1, 2])
def test_other_iteration_counts_match_oracle(maxiter, seed):
    """The early spectral steps are solved to a relaxed tolerance that depends on maxiter (solver.RotationSolver.run):
    rotations against the oracle for other iteration counts than the goldens' 4 (probe over maxiter 1..8 x 40 seeds:
    <= 1.3e-9 rad in f64, 2.8e-6 rad in f32)."""
    from oracle import bipgo_oracle as orc
    from vican.bipgo import bipartite_se3sync
    mode, scene, flat, (wr, wt), filt, dt = make_case(seed)
    assert mode == "camera"
    src = synth.edges_to_dict(flat, SE3)
    cons = synth.constraints_from_scene(scene, SE3)
    nr, nt, ff = gc.CALLABLES[wr], gc.CALLABLES[wt], gc.CALLABLES[filt]
    res = bipartite_se3sync(src, cons, nr, nt, ff, maxiter, "conjugate_gradient", dt)
    ref = orc.bipartite_se3sync(src, cons, nr, nt, ff, maxiter, "conjugate_gradient", dt, loop=True)
    R = np.stack([np.asarray(res[k].R(), dtype=np.float64) for k in res])
    Rr = np.stack([np.asarray(ref[k].R(), dtype=np.float64) for k in ref])
    assert float(geodesic(R, Rr).max()) < (1e-7 if dt == np.float64 else 5e-6)
