"""G9: parity at BASELINE.json configs[2] scale (340 cameras x 10000 timesteps, 80000 source edges) against outputs of
the REAL reference (tests/golden/g9_large_shop.npz; reference wall-clock there: ~15 s).  Inputs are regenerated from
the seeded description in golden_cases.LARGE_SHOP and checked against the digest stored with the outputs."""
import numpy as np
import pytest

import golden_cases as gc
from util import expected, load_golden, pose_errors, translation_tol
from vican_amd import synth
from vican_amd.geometry import SE3


def digest(flat):
    return np.array([float(np.sum(flat["R"] * np.arange(1, flat["R"].size + 1).reshape(flat["R"].shape) % 7)),
                     float(np.sum(flat["t"])), float(np.sum(flat["corners"])), float(len(flat["cam_key"]))])


@pytest.fixture(scope="module")
def case():
    g = load_golden("g9_large_shop")
    scene, flat = gc.build_flat(gc.LARGE_SHOP)
    assert np.array_equal(digest(flat), g["digest"]), "regenerated inputs differ from the ones the reference was run on"
    src = synth.edges_to_dict(flat, SE3)
    cons = synth.constraints_from_scene(scene, SE3)
    fns = tuple(gc.CALLABLES[gc.LARGE_SHOP[k]] for k in ("noise_r", "noise_t", "filt"))
    return g, src, cons, fns


def test_oracle_matches_reference_at_large_shop_scale(case):
    from oracle import bipgo_oracle as orc
    g, src, cons, (nr, nt, ff) = case
    exp = expected(g, "conjugate_gradient", "float32")
    info = {}
    res = orc.bipartite_se3sync(src, cons, nr, nt, ff, gc.MAXITER, "conjugate_gradient", np.float32, loop=True, info=info)
    rot, tr = pose_errors(res, exp)
    assert rot < 5e-6, rot
    # 105 CG iterations at relres 1e-5 leave the reference 16 m from the converged solution of its own system
    # (golden `dist_tight`); the restatement makes the same scipy calls on the same data and still lands close
    assert tr < translation_tol(exp, False), tr
    assert abs(info["cg_iters"] - int(exp["cg_iters"])) <= 2


@pytest.mark.gpu
def test_dropin_matches_reference_at_large_shop_scale(case):
    from vican.bipgo import bipartite_se3sync
    g, src, cons, (nr, nt, ff) = case
    exp = expected(g, "conjugate_gradient", "float32")
    info = {}
    res = bipartite_se3sync(src, constraints=cons, noise_model_r=nr, noise_model_t=nt, edge_filter=ff, maxiter=gc.MAXITER,
                            lsqr_solver="conjugate_gradient", dtype=np.float32, info=info)
    rot, tr = pose_errors(res, exp)
    print("large_shop scale: rot %.2e rad, trans %.3e m (reference is %.1f m from its own converged solution), cg %d vs %d, "
          "solve %.1f ms (reference %.1f s)" % (rot, tr, float(exp["dist_tight"]), info["cg_iters"], int(exp["cg_iters"]),
                                               1e3 * (info["t_rot"] + info["t_trans"]), float(exp["ref_wall_s"])))
    assert rot < 5e-6 <= 1e-4, rot
    assert tr < translation_tol(exp, False), tr
    # scipy's stopping rule is reproduced, but after ~100 iterations without re-orthogonalisation the residual hovers
    # around rtol |b| non-monotonically (as on g4, tests/test_parity_gpu.py): which dip below the threshold is caught
    # first moves by tens of iterations under rounding-level differences (measured: 118 against the reference's 105)
    assert abs(info["cg_iters"] - int(exp["cg_iters"])) <= 30
    ev3 = np.sort(info["evals"][:, :3], axis=1)
    evr = np.sort(exp["evals"], axis=1)[:, :3]
    assert np.abs(ev3 - evr).max() < 1e-4 * np.abs(exp["evals"]).max()
