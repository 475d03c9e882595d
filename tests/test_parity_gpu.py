"""End-to-end parity on the GPU: the drop-in API (edge dict in, pose dict out) through the
HIP kernels, against outputs of the REAL reference (tests/golden/*.npz)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import golden_cases as gc                                                   # noqa: E402
from util import e2e_translation_tol, expected, iteration_slack, load_golden, pose_errors, rebuild_inputs, translation_tol   # noqa: E402

CG_RUNS = [(n, d) for n, c in gc.CASES.items() for (s, d) in c["runs"] if s == "conjugate_gradient"]

# SE(3) parity tolerance stated by BASELINE.json's north star: 1e-4 rad / 1e-4 m
ROT_TOL = {"float64": 1e-7, "float32": 5e-6}


@pytest.mark.parametrize("name,dt", CG_RUNS)
def test_dropin_api_matches_reference(name, dt):
    from vican.bipgo import bipartite_se3sync, object_bipartite_se3sync      # the shim import path
    g = load_golden(name)
    case, src, cons, (nr, nt, ff) = rebuild_inputs(name, g)
    exp = expected(g, "conjugate_gradient", dt)
    dtype = np.dtype(dt).type
    info = {}
    if case["mode"] == "camera":
        res = bipartite_se3sync(src, constraints=cons, noise_model_r=nr, noise_model_t=nt, edge_filter=ff,
                                maxiter=gc.MAXITER, lsqr_solver="conjugate_gradient", dtype=dtype, info=info)
    else:
        res = object_bipartite_se3sync(src, noise_model_r=nr, noise_model_t=nt, edge_filter=ff,
                                       maxiter=gc.MAXITER, lsqr_solver="conjugate_gradient", dtype=dtype, info=info)
    rot, tr = pose_errors(res, exp)
    from conftest import record_parity
    record_parity(name, dt, "drop-in", rot, tr, e2e_translation_tol(name, dt), info["cg_iters"], int(exp["cg_iters"]))
    assert rot < ROT_TOL[dt] <= 1e-4, rot
    assert tr < e2e_translation_tol(name, dt), tr
    if name == "g4_illcond" and info["cg_iters"] == int(exp["cg_iters"]):
        assert tr < 1e-3, tr          # stopped at the reference's iteration: then the iterate itself must match
    # iteration count: exact (+-1) on the well-conditioned cases.  On the heavy-tailed-weight case g4 (the
    # reference itself is 17 m from the converged solution there) CG has lost conjugacy after ~15 iterations
    # and the relative residual jumps erratically between 4e-4 and 5e-6: it dips below rtol = 1e-5 at
    # iterations 21 (9.0e-6, marginal), 25, 28 and 31/32, and which dip is caught first flips under 1e-14
    # perturbations of the rotations (measured with the NumPy backend: 20 or 24 iterations) - so only the
    # window of those dips is comparable there.
    slack = iteration_slack(name, dt)
    assert abs(info["cg_iters"] - int(exp["cg_iters"])) <= slack
    first = next(iter(res.values()))
    assert first.R().dtype == dtype and first.t().dtype == np.float64       # bipgo.py:484-487 types
    ev3 = np.sort(info["evals"][:, :3], axis=1)
    evr = np.sort(exp["evals"], axis=1)[:, :3]
    assert np.abs(ev3 - evr).max() < (1e-7 if dt == "float64" else 1e-4) * np.abs(exp["evals"]).max()


@pytest.mark.parametrize("name,dt,tol", [("g2_small", "float64", 1e-8), ("g3_medium", "float64", 1e-6), ("g3_medium", "float32", 2e-3),
                                         ("g4_illcond", "float64", 3e-3)])   # g4: cond ~1e7 x relres 1e-10 (reference: 17 m)
def test_tight_mode_on_gpu(name, dt, tol):
    """tight=True: translations converged to relres 1e-10 (Jacobi-scaled CG on the device) equal the converged
    solution of the reference's own system (`t_tight` golden), rotations unchanged."""
    from vican.bipgo import bipartite_se3sync
    g = load_golden(name)
    case, src, cons, (nr, nt, ff) = rebuild_inputs(name, g)
    exp = expected(g, "conjugate_gradient", dt)
    info = {}
    res = bipartite_se3sync(src, constraints=cons, noise_model_r=nr, noise_model_t=nt, edge_filter=ff, maxiter=gc.MAXITER,
                            lsqr_solver="conjugate_gradient", dtype=np.dtype(dt).type, info=info, tight=True)
    rot, _ = pose_errors(res, exp)
    assert rot < ROT_TOL[dt]
    t = np.stack([np.asarray(v.t(), dtype=np.float64) for v in res.values()])
    # the golden's scipy run to 1e-14 drifts along the null space (a common shift of all nodes) when the f32
    # incidence matrix makes the system only nearly singular: compare in the zero-sum gauge
    tt = exp["t_tight"] - exp["t_tight"].mean(0)
    assert float(np.linalg.norm(t - tt, axis=1).max()) < tol
    assert info["cg_relres"] < 1e-9 and abs(t.sum(0)).max() < 1e-7 * max(1.0, np.abs(t).max())
    with pytest.raises(UnboundLocalError):                                    # the solver string is still validated
        bipartite_se3sync(src, cons, nr, nt, ff, 4, "cholesky", np.dtype(dt).type, tight=True)


def test_array_entry_point_is_bit_identical_to_the_dict_api():
    """bipartite_se3sync_arrays (detections as arrays, no per-edge callables) returns exactly what the drop-in returns
    for the same kept edges and weights."""
    from vican_amd.bipgo import bipartite_se3sync, bipartite_se3sync_arrays
    g = load_golden("g3_medium")
    case, src, cons, (nr, nt, ff) = rebuild_inputs("g3_medium", g)
    a = bipartite_se3sync(src, cons, nr, nt, ff, gc.MAXITER, "conjugate_gradient", np.float32)
    kept = [(k, v) for k, v in src.items() if ff(v)]
    R = np.stack([np.asarray(v["pose"].R(), dtype=np.float64) for _, v in kept])
    t = np.stack([np.asarray(v["pose"].t(), dtype=np.float64).reshape(3) for _, v in kept])
    info = {}
    b = bipartite_se3sync_arrays([k[0] for k, _ in kept], [k[1].split("_")[0] for k, _ in kept], [k[1].split("_")[1] for k, _ in kept],
                                 R, t, np.array([nr(v) for _, v in kept]), np.array([nt(v) for _, v in kept]), cons, gc.MAXITER,
                                 "conjugate_gradient", np.float32, info=info)
    assert list(a) == list(b) and info["cg_iters"] > 0
    for k in a:
        assert np.array_equal(a[k].R(), b[k].R()) and np.array_equal(a[k].t(), b[k].t())


def test_gauge_and_error_behaviour():
    from vican_amd.bipgo import bipartite_se3sync
    g = load_golden("g2_small")
    case, src, cons, (nr, nt, ff) = rebuild_inputs("g2_small", g)
    out = bipartite_se3sync(src, cons, nr, nt, ff, 4, "conjugate_gradient", np.float64)
    first_cam = sorted(k for k in out if "_" not in k)[0]
    assert np.abs(out[first_cam].R() - np.eye(3)).max() < 1e-9              # gauge: first camera = identity
    assert np.abs(sum(p.t() for p in out.values())).max() < 1e-8             # CG from 0: translations sum to 0
    with pytest.raises(KeyError):                                             # unknown marker id (bipgo.py:209)
        bad = dict(cons); bad.pop(sorted(bad)[-1])
        bipartite_se3sync(src, bad, nr, nt, ff, 4, "conjugate_gradient", np.float64)
    with pytest.raises(UnboundLocalError):                                    # bipgo.py:476-487
        bipartite_se3sync(src, cons, nr, nt, ff, 4, "cholesky", np.float64)


def test_roundtrip_property_at_scale():
    """Size-independent property on a large_shop-like graph (no oracle at this size): with
    noise-free measurements the solver must return the ground truth up to gauge."""
    import torch
    from vican_amd import synth
    from vican_amd.device import HipBackend, LocalGraph
    from vican_amd.geometry import geodesic
    from vican_amd.solver import Comm, solve_on_backend
    C, T = 340, 10000
    dev = torch.device("cuda:0")
    gr = synth.make_merged_graph_torch(C, T, 4, dev, torch.float64, seed=5, sigma_r=0.0, sigma_t=0.0)
    g = LocalGraph(C, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"])
    rc, Rt, x_c, x_t, st = solve_on_backend(HipBackend(g), Comm(), 4, 3 * (C + T), rtol=1e-12)
    Rc = rc.reshape(C, 3, 3).transpose(1, 2).cpu().numpy()
    Rgt = gr["R_cam"].cpu().numpy()
    G = Rgt[0]                                     # our camera 0 is the identity
    assert geodesic(G @ Rc, Rgt).max() < 1e-9
    Rtt = Rt.reshape(T, 3, 3).transpose(1, 2).cpu().numpy()
    assert geodesic(G @ Rtt, gr["R_obj"].cpu().numpy()).max() < 1e-9
    pc, pgt = x_c.cpu().numpy() @ G.T, gr["p_cam"].cpu().numpy()
    assert np.abs((pc - pc.mean(0)) - (pgt - pgt.mean(0))).max() < 1e-6


def test_full_size_stress_graph_properties():
    """BASELINE.json's stress configuration at FULL size (1000 cameras x 100 000 timesteps x 250 cameras per
    timestep = 25 M merged edges, 1 GB of f32 blocks - far beyond what the oracle can run), checked through
    size-independent properties: (1) noise-free measurements -> the ground truth up to gauge, to f32 accuracy;
    (2) the operator sweep is bit-reproducible although chunks are handed to workgroups dynamically;
    (3) the result does not depend on the workgroup count / chunk assignment."""
    import torch
    from vican_amd import synth
    from vican_amd.device import HipBackend, LocalGraph
    from vican_amd.geometry import geodesic
    from vican_amd.solver import Comm, solve_on_backend
    C, T = 1000, 100000
    dev = torch.device("cuda:0")
    gr = synth.make_merged_graph_torch(C, T, 250, dev, torch.float32, seed=9, sigma_r=0.0, sigma_t=0.0)
    g = LocalGraph(C, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"])
    assert g.n_edges == 25_000_000
    K = HipBackend(g)
    rc, Rt, x_c, x_t, st = solve_on_backend(K, Comm(), 4, 3 * (C + T))
    Rc = rc.reshape(C, 3, 3).transpose(1, 2).cpu().numpy()
    Rgt = gr["R_cam"].cpu().numpy()
    G = Rgt[0]
    assert geodesic(G @ Rc, Rgt).max() < 2e-6                       # f32 blocks: 6e-8 relative rounding per entry
    Rtt = Rt.reshape(T, 3, 3).transpose(1, 2).cpu().numpy()
    assert geodesic(G @ Rtt, gr["R_obj"].cpu().numpy()).max() < 2e-6
    pc, pgt = x_c.cpu().numpy() @ G.T, gr["p_cam"].cpu().numpy()
    assert np.abs((pc - pc.mean(0)) - (pgt - pgt.mean(0))).max() < 5e-4   # scipy-style CG at rtol 1e-5
    assert st["sweeps"] <= 24 and st["converged"]
    # (2) same sweep twice: identical bits
    lamT, cd = K.empty(T, 9), K.empty(C)
    K.init_duals(lamT, cd)
    x = torch.linalg.qr(torch.randn(3 * C, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1)))[0].contiguous()
    z1, z2 = K.zeros(3 * C, 3), K.zeros(3 * C, 3)
    K.block_op(lamT, x, z1); K.block_op(lamT, x, z2)
    assert torch.equal(z1, z2)
    # (3) a different workgroup count (other chunk-to-workgroup assignment, other slab count): exact integer sums ->
    # identical up to the fixed-point scale (n_add differs), i.e. to ~2^-40 of the bound
    g2 = LocalGraph(C, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"], n_wg=97)
    K2 = HipBackend(g2)
    lam2, cd2 = K2.empty(T, 9), K2.empty(C)
    K2.init_duals(lam2, cd2)
    z3 = K2.zeros(3 * C, 3)
    K2.block_op(lam2, x, z3)
    assert float((z3 - z1).abs().max()) < 1e-9 * float(z1.abs().max())


@pytest.mark.parametrize("C,T,k,mode", [(1000, 6000, 250, "dense rows: wave-per-row phase 2"),
                                        (340, 10000, 4, "sparse rows: row-parallel phase 2"),
                                        (64, 3000, 20, "mid")])
def test_sweeps_are_bit_reproducible(C, T, k, mode):
    """Operator and dual-update sweeps repeated on the same input give identical bits although the chunks are
    handed to the workgroups dynamically (exact integer accumulation) - for both phase-2 variants."""
    import torch
    from vican_amd import synth
    from vican_amd.device import HipBackend, LocalGraph
    dev = torch.device("cuda:0")
    gr = synth.make_merged_graph_torch(C, T, k, dev, torch.float32, seed=3)
    g = LocalGraph(C, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"])
    K = HipBackend(g)
    lamT, cd = K.empty(T, 9), K.empty(C)
    K.init_duals(lamT, cd)
    x = torch.linalg.qr(torch.randn(3 * C, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1)))[0].contiguous()
    z0 = K.zeros(3 * C, 3)
    K.block_op(lamT, x, z0)
    for _ in range(25):
        z = K.zeros(3 * C, 3)
        K.block_op(lamT, x, z)
        assert torch.equal(z, z0)
    fx0 = g.fx.clone()
    Rt0, lam0 = K.zeros(T, 9), lamT.clone()
    K.dual_update(x, Rt0, lam0)
    for _ in range(10):
        g.fx.copy_(fx0)
        Rt, lam = K.zeros(T, 9), lamT.clone()
        K.dual_update(x, Rt, lam)
        assert torch.equal(Rt, Rt0) and torch.equal(lam, lam0)


def test_solve_is_bit_reproducible():
    """All edge-side sums are 64-bit fixed point (integer atomics) and all camera-side reductions
    have a fixed order, so two solves of the same problem - including a fresh pack of the graph -
    return bit-identical poses (the reference's loose CG would amplify summation-order noise)."""
    from vican_amd.bipgo import bipartite_se3sync
    g = load_golden("g3_medium")
    case, src, cons, (nr, nt, ff) = rebuild_inputs("g3_medium", g)
    outs = []
    for _ in range(3):
        res = bipartite_se3sync(src, cons, nr, nt, ff, gc.MAXITER, "conjugate_gradient", np.float64)
        outs.append((np.stack([p.R() for p in res.values()]), np.stack([p.t() for p in res.values()])))
    for R, t in outs[1:]:
        assert np.array_equal(R, outs[0][0]) and np.array_equal(t, outs[0][1])


LSQR_RUNS = [(n, d) for n, c in gc.CASES.items() for (s, d) in c["runs"] if s == "direct"]


@pytest.mark.parametrize("name,dt", LSQR_RUNS)
def test_dropin_direct_lsqr_matches_reference(name, dt):
    """lsqr_solver="direct" (scipy LSQR in the reference, bipgo.py:479-480) through the HIP LSQR kernels."""
    from vican.bipgo import bipartite_se3sync
    g = load_golden(name)
    case, src, cons, (nr, nt, ff) = rebuild_inputs(name, g)
    exp = expected(g, "direct", dt)
    info = {}
    res = bipartite_se3sync(src, constraints=cons, noise_model_r=nr, noise_model_t=nt, edge_filter=ff,
                            maxiter=gc.MAXITER, lsqr_solver="direct", dtype=np.dtype(dt).type, info=info)
    rot, tr = pose_errors(res, exp)
    from conftest import record_parity
    record_parity(name, dt, "drop-in LSQR", rot, tr, 2e-6 if dt == "float64" else 5e-4, info["lsqr_iters"], None)
    assert rot < ROT_TOL[dt]
    assert tr < (2e-6 if dt == "float64" else 5e-4), tr
    assert info["lsqr_istop"] in (1, 2) and info["lsqr_iters"] > 0
