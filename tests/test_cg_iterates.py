"""Iteration-matched translation parity against the REAL reference's CG iterates - and what it shows.

The reference's translations are a LOOSELY converged scipy-CG iterate (rtol 1e-5 on a singular Laplacian system,
bipgo.py:476-478).  With non-unit weights which iteration a run stops at is a coin the reference itself flips (g9: 101, 103
and 105 in three runs of the real reference), and one iteration moves the iterate by 1e-4 .. 8e-4 m there - so the end-to-end
tests bound translations by the reference's own reproducibility instead of the north star's 1e-4 m.  This file compares
ITERATE k of the product with ITERATE k of the reference, which takes the stopping coin out: golden ``g10_cg_iterates.npz``
(tests/golden/make_golden.py: cg_iterates_case) holds the reference's iterates x_k - scipy's callback on its one cg call -
for k = 10, 25, 50, 75 and the last six iterations of each run of g3 and g9, and the product is told to run exactly k
iterations (``cg_stop_at`` / ``TranslationSolver.solve(stop_at=)``).

What the golden ALSO holds is the reference's own iterate-matched reproducibility (``self_dx``: its cg call repeated eight
times on right-hand sides perturbed by 1e-15 relative, forced to the same iteration count, distance of iterate k to iterate
k).  It settles what can be asserted: the trajectory itself is chaotic, not just its stopping point.  On g9 the reference's
own iterate 10 is reproducible to 2e-7 m, iterate 25 moves by 0.15 - 0.45 METRES (a Ritz value converges one iteration
earlier or later), iterate 75 by 4e-3 m, and every iterate of the stopping window (98..105) by 2.5e-4 .. 7.7e-4 m; on g3 by
1e-4 .. 1.3e-3 m in the window - in float64 and float32 alike.  No implementation - scipy itself included - reproduces
iterate k >= 25 of these systems to 1e-4 m.  Asserted here: (i) before the chaos sets in (k = 10) the product reproduces
the reference's iterate to 1e-6 m - right-hand side, Laplacian product and recurrence are the reference's; (ii) at every
stored k the product stays inside 4 x the reference's own movement OF THAT ITERATE; (iii) the same through the drop-in call
at the iteration where that run of the reference stopped.

``g11_unit_scale.npz``: UNIT-weight scenes of large_shop and small_room size, where scipy's CG takes 15-17 iterations and
is reproducible to rounding: the north star's 1e-4 m holds with two orders to spare (1e-6 m in float64).
"""
import numpy as np
import pytest
import torch

import golden_cases as gc
from vican_amd.backend_cpu import NumpyBackend
from util import load_golden
from vican_amd import frontend, synth
from vican_amd.geometry import SE3, geodesic
from vican_amd.solver import Comm, TranslationSolver

NORTH_STAR_M = 1e-4


def iterate_tol(e, i):
    """Bound for stored iterate i of a run: 4 x the largest movement of the reference's OWN iterates within three iterations
    of it (8 trials each under 1e-15 perturbations: a small sample of a heavy-tailed distribution - in g9's stopping window
    the per-iterate maxima range from 2.5e-4 to 7.7e-4 m - so neighbouring iterates share their evidence), floored at 1e-6 m
    where the reference is reproducible to rounding."""
    k = e["k"].astype(np.int64)
    near = np.abs(k - k[i]) <= 3
    return max(1e-6, 4.0 * float(e["self_dx"][:, near].max()))


RUNS = [(n, dt) for n, c in gc.ITERATE_CASES.items() for s, dt in c["runs"] if s == "conjugate_gradient"]


def digest(flat):
    return np.array([float(np.sum(flat["R"] * np.arange(1, flat["R"].size + 1).reshape(flat["R"].shape) % 7)),
                     float(np.sum(flat["t"])), float(np.sum(flat["corners"])), float(len(flat["cam_key"]))])


_CACHE = {}


def inputs(case_name, case):
    """Regenerated source edges of a seeded case (checked against the digest stored with the reference's outputs)."""
    if case_name not in _CACHE:
        scene, flat = gc.build_flat(case)
        src = synth.edges_to_dict(flat, SE3)
        cons = synth.constraints_from_scene(scene, SE3)
        fns = tuple(gc.CALLABLES[case[k]] for k in ("noise_r", "noise_t", "filt"))
        _CACHE[case_name] = (digest(flat), src, cons, fns)
    return _CACHE[case_name]


def iterate_golden(name, dt):
    g = load_golden("g10_cg_iterates")
    tag = "%s_%s_" % (name, dt)
    e = {k[len(tag):]: v for k, v in g.items() if k.startswith(tag)}
    d, src, cons, fns = inputs(name, gc.ITERATE_CASES[name])
    assert np.array_equal(d, e["digest"]), "regenerated inputs differ from the ones the reference was run on"
    e["x"] = {int(k): e["t"] + e["dx"][i].astype(np.float64) for i, k in enumerate(e["k"])}     # iterate k, node order = key order
    return e, src, cons, fns


def reference_rotations(prob, e):
    R = {str(k): np.asarray(e["R"][i], dtype=np.float64) for i, k in enumerate(e["keys"])}
    rc = np.stack([R[str(c)].T for c in prob.cam_names]).reshape(-1, 3)
    rt = np.stack([R[str(s) + "_0"].T for s in prob.time_names]).reshape(-1, 9)
    return rc, rt


def stage_iterate(K, prob, e, k):
    """Translation stage alone (the reference's rotations in), exactly k iterations -> positions in the reference's key order."""
    rc, rt = reference_rotations(prob, e)
    tr = TranslationSolver(K, Comm.single())
    tr.setup(K.from_numpy(rc), K.from_numpy(rt))
    x_c, x_t = tr.solve(3 * (prob.n_cam + prob.n_time), stop_at=k)
    assert tr.info["cg_iters"] == k
    pos = {str(c): x_c.cpu().numpy()[i] for i, c in enumerate(prob.cam_names)}
    pos.update({str(s) + "_0": x_t.cpu().numpy()[i] for i, s in enumerate(prob.time_names)})
    return np.stack([pos[str(kk)] for kk in e["keys"]])


def dist(a, b):
    return float(np.linalg.norm(a - b, axis=1).max())


# ---------------------------------------------------------------------------------------------- CPU: oracle + host logic

@pytest.mark.parametrize("name,dt", [r for r in RUNS if r[0] == "g3_medium"])
def test_oracle_iterates_match_the_reference(name, dt):
    """The oracle makes the same scipy call on the same system: fed the reference's rotations, its iterates ARE the
    reference's (every stored k, to rounding) - the restatement is pinned at the level of single CG iterations."""
    from oracle import bipgo_oracle as orc
    e, src, cons, (nr, nt, ff) = iterate_golden(name, dt)
    flat = orc.flatten_edges(src, cons, nr, nt, ff)
    R = {str(k): np.asarray(e["R"][i], dtype=np.float64) for i, k in enumerate(e["keys"])}
    Rc = np.stack([R[c] for c in flat["cam_names"]]).astype(np.dtype(dt))
    Rt = np.stack([R[t + "_0"] for t in flat["time_names"]]).astype(np.dtype(dt))
    its = {}
    cg0 = orc.cg

    def cg(A, b, *a, **kw):
        n = [0]

        def cb(xk):
            n[0] += 1
            its[n[0]] = np.array(xk).reshape(-1, 3)
        kw.pop("callback", None)
        return cg0(A, b, *a, callback=cb, **kw)
    orc.cg = cg
    try:
        orc.translation_arrays(len(flat["tnodes"]), flat["tnode_of_cam"][flat["cam_idx"]], flat["tnode_of_time"][flat["time_idx"]],
                               Rc[flat["cam_idx"]], Rt[flat["time_idx"]], flat["t"], flat["rel_R"], flat["rel_t"], flat["k_t"],
                               "conjugate_gradient", np.dtype(dt).type, {}, False)
    finally:
        orc.cg = cg0
    assert [str(n) for n in flat["tnodes"]] == [str(k) for k in e["keys"]]
    assert len(its) == int(e["cg_iters"])
    for i, k in enumerate(e["k"]):
        # same scipy call, right-hand side summed in another order (1e-16): inside the reference's own movement of iterate k
        # (+ the float32 resolution of the stored differences: 1.2e-7 of the distance still to go)
        tol = iterate_tol(e, i) + 2.5e-7 * float(np.abs(e["dx"][i]).max())
        assert dist(its[int(k)], e["x"][int(k)]) < tol, (k, dist(its[int(k)], e["x"][int(k)]), tol)
    assert dist(its[10], e["x"][10]) < 1e-6


def test_the_references_own_iterates_are_not_reproducible_to_the_north_star():
    """The golden's evidence, asserted so that it cannot silently change: under 1e-15 perturbations of its right-hand side the
    REAL reference's iterate 10 is reproducible to < 1e-6 m on both cases, while in the stopping window every g9 iterate and
    most g3 iterates move by more than the north star's 1e-4 m; g9's iterate 25 moves by more than 0.1 m."""
    g = load_golden("g10_cg_iterates")
    for name, dt in RUNS:
        tag = "%s_%s_" % (name, dt)
        k, mv = [int(x) for x in g[tag + "k"]], g[tag + "self_dx"].max(0)
        assert mv[k.index(10)] < 1e-6
        window = mv[-gc.ITERATE_LAST:-1]                           # (the last stored iterate is the converged-by-rtol one)
        if name == "g9_large_shop":
            assert window.min() > NORTH_STAR_M and mv[k.index(25)] > 0.1, (name, dt, mv)
        else:
            assert np.median(window) > NORTH_STAR_M, (name, dt, mv)


@pytest.mark.parametrize("name,dt", [r for r in RUNS if r[0] == "g3_medium"])
def test_host_logic_reproduces_iterate_k(name, dt):
    """TranslationSolver(stop_at=k) on the NumPy stand-in backend: the host state machine runs exactly k iterations of
    scipy's recurrence (k updates of x), for every stored k."""
    e, src, cons, (nr, nt, ff) = iterate_golden(name, dt)
    prob = frontend.flatten(src, cons, nr, nt, ff, np.dtype(dt).type)
    K = NumpyBackend(prob.n_cam, prob.row_ptr, prob.col, prob.blk, prob.a, prob.w, prob.u, prob.v, storage=np.dtype(dt).type,
                     deg_t=prob.deg_t, deg_c=prob.deg_c)
    for i, k in enumerate(e["k"]):
        d = dist(stage_iterate(K, prob, e, int(k)), e["x"][int(k)])
        assert d < iterate_tol(e, i), (k, d, iterate_tol(e, i))
    assert dist(stage_iterate(K, prob, e, 10), e["x"][10]) < 1e-6


# ---------------------------------------------------------------------------------------------- GPU

def hip_backend(prob, dt):
    from vican_amd.device import HipBackend, LocalGraph
    dev = torch.device("cuda", torch.cuda.current_device())
    tdt = torch.float32 if dt == "float32" else torch.float64
    to = lambda a, d=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev, d)
    g = LocalGraph(prob.n_cam, to(prob.row_ptr, torch.int32), to(prob.col, torch.int32), to(prob.blk, tdt), to(prob.a, tdt),
                   to(prob.w), to(prob.u), to(prob.v), deg_t=to(prob.deg_t), deg_c=to(prob.deg_c))
    return HipBackend(g)


@pytest.mark.gpu
@pytest.mark.parametrize("name,dt", RUNS)
def test_translation_kernels_reproduce_every_stored_iterate(name, dt):
    """Right-hand side, Laplacian product and CG recurrence on the GPU, the reference's rotations fed in: iterate k of the
    product against iterate k of the reference for every stored k - 1e-6 m at k = 10 (before the trajectory turns chaotic),
    inside 4 x the reference's own movement of that iterate everywhere."""
    e, src, cons, (nr, nt, ff) = iterate_golden(name, dt)
    prob = frontend.flatten(src, cons, nr, nt, ff, np.dtype(dt).type)
    K = hip_backend(prob, dt)
    got = {int(k): dist(stage_iterate(K, prob, e, int(k)), e["x"][int(k)]) for k in e["k"]}
    print("%s %s: product iterate k vs reference iterate k, m (the reference's own movement of iterate k): %s" % (
        name, dt, " ".join("%d:%.1e(%.1e)" % (k, got[int(k)], e["self_dx"][:, i].max()) for i, k in enumerate(e["k"]))))
    assert got[10] < 1e-6, got
    for i, k in enumerate(e["k"]):
        assert got[int(k)] < iterate_tol(e, i), (k, got[int(k)], iterate_tol(e, i))


@pytest.mark.gpu
@pytest.mark.parametrize("name,dt", RUNS)
def test_dropin_at_the_references_stopping_iteration(name, dt):
    """End to end through the drop-in call (own rotations), told to stop where THIS run of the reference stopped: rotations
    inside 1e-7 / 5e-6 rad, translations inside 4 x the reference's own movement of that iterate (the two eigen-solvers'
    rotations differ by 1e-9 rad in float64 and ~3e-7 rad in float32 - perturbations of the right-hand side far above the
    1e-15 of the yardstick, hence the 2e-3 m ceiling the other end-to-end tests use as well)."""
    from vican.bipgo import bipartite_se3sync
    e, src, cons, (nr, nt, ff) = iterate_golden(name, dt)
    k = int(e["cg_iters"])
    info = {}
    res = bipartite_se3sync(src, constraints=cons, noise_model_r=nr, noise_model_t=nt, edge_filter=ff, maxiter=gc.MAXITER,
                            lsqr_solver="conjugate_gradient", dtype=np.dtype(dt).type, info=info, cg_stop_at=k)
    assert [str(x) for x in res.keys()] == [str(x) for x in e["keys"]]
    R = np.stack([np.asarray(v.R(), dtype=np.float64) for v in res.values()])
    t = np.stack([np.asarray(v.t(), dtype=np.float64) for v in res.values()])
    rot, tr = float(geodesic(R, np.asarray(e["R"], dtype=np.float64)).max()), dist(t, e["t"])
    print("%s %s: drop-in stopped at the reference's iteration %d: rot %.2e rad, trans %.2e m" % (name, dt, k, rot, tr))
    assert info["cg_iters"] == k
    assert rot < (5e-6 if dt == "float32" else 1e-7), rot
    assert tr < min(max(iterate_tol(e, len(e["k"]) - 1), 4.0 * float(e["self_dx"].max(0)[-gc.ITERATE_LAST:].max())), 2e-3), tr


UNIT_RUNS = [(n, dt) for n, c in gc.UNIT_SCALE.items() for s, dt in c["runs"]]


def unit_golden(name, dt):
    g = load_golden("g11_unit_scale")
    tag = "%s_%s_" % (name, dt)
    e = {k[len(tag):]: v for k, v in g.items() if k.startswith(tag)}
    d, src, cons, fns = inputs(name, gc.UNIT_SCALE[name])
    assert np.array_equal(d, g[name + "_digest"]), "regenerated inputs differ from the ones the reference was run on"
    return e, src, cons, fns


@pytest.mark.parametrize("name,dt", [r for r in UNIT_RUNS if r[0] == "unit_small_room_2000"])
def test_oracle_matches_reference_on_unit_weight_scene(name, dt):
    from oracle import bipgo_oracle as orc
    e, src, cons, (nr, nt, ff) = unit_golden(name, dt)
    info = {}
    res = orc.bipartite_se3sync(src, cons, nr, nt, ff, gc.MAXITER, "conjugate_gradient", np.dtype(dt).type, info=info)
    assert [str(x) for x in res.keys()] == [str(x) for x in e["keys"]]
    R = np.stack([np.asarray(v.R(), dtype=np.float64) for v in res.values()])
    t = np.stack([np.asarray(v.t(), dtype=np.float64) for v in res.values()])
    assert float(geodesic(R, np.asarray(e["R"], dtype=np.float64)).max()) < (5e-6 if dt == "float32" else 1e-8)
    assert dist(t, e["t"]) < (1e-5 if dt == "float32" else 1e-7)
    assert info["cg_iters"] == int(e["cg_iters"])


@pytest.mark.gpu
@pytest.mark.parametrize("name,dt", UNIT_RUNS)
def test_dropin_matches_reference_on_unit_weight_scenes_at_dataset_size(name, dt):
    """large_shop (340 x 10000) and small_room (40 x 2000 / 5000) sized scenes with unit weights: scipy's CG takes 15-17
    iterations and is reproducible to rounding, so the drop-in call is compared with the reference's output directly -
    translations to 1e-6 m in float64 (100x inside the north star) and, in float32 - where the two eigen-solvers' rotations
    differ by ~1e-7 rad, which moves a right-hand side built from 10 m lever arms by micrometres - to 2e-5 m."""
    from vican.bipgo import bipartite_se3sync
    e, src, cons, (nr, nt, ff) = unit_golden(name, dt)
    info = {}
    res = bipartite_se3sync(src, constraints=cons, noise_model_r=nr, noise_model_t=nt, edge_filter=ff, maxiter=gc.MAXITER,
                            lsqr_solver="conjugate_gradient", dtype=np.dtype(dt).type, info=info)
    assert [str(x) for x in res.keys()] == [str(x) for x in e["keys"]]
    R = np.stack([np.asarray(v.R(), dtype=np.float64) for v in res.values()])
    t = np.stack([np.asarray(v.t(), dtype=np.float64) for v in res.values()])
    rot, tr = float(geodesic(R, np.asarray(e["R"], dtype=np.float64)).max()), dist(t, e["t"])
    print("%s %s: rot %.2e rad, trans %.2e m, cg %d vs %d (reference %.1f s, drop-in solve %.1f ms)" % (
        name, dt, rot, tr, info["cg_iters"], int(e["cg_iters"]), float(e["ref_wall_s"]), 1e3 * (info["t_rot"] + info["t_trans"])))
    from conftest import record_parity
    record_parity("g11_" + name, dt, "drop-in", rot, tr, 2e-5 if dt == "float32" else 1e-6, info["cg_iters"], int(e["cg_iters"]))
    assert rot < (5e-6 if dt == "float32" else 1e-7), rot
    assert tr < (2e-5 if dt == "float32" else 1e-6), tr
    assert info["cg_iters"] == int(e["cg_iters"])
    ev3 = np.sort(info["evals"][:, :3], axis=1)
    evr = np.sort(e["evals"], axis=1)[:, :3]
    assert np.abs(ev3 - evr).max() < (1e-4 if dt == "float32" else 1e-7) * np.abs(e["evals"]).max()
