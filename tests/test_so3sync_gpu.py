"""SURVEY 8(f) row 4 on the GPU: the one-pass bipartite operator (sweep MODE 2), the polar step without det fix,
and the drop-in `bipartite_so3sync` against outputs of the REAL reference (tests/golden/g8_so3sync.npz)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import golden_cases as gc                                     # noqa: E402
from vican_amd.backend_cpu import svd_polar                           # noqa: E402
from test_kernels_gpu import CONFIGS, make_backends           # noqa: E402
from test_so3sync_cpu import CASES, golden, inputs            # noqa: E402


@pytest.mark.parametrize("cfg", [c for c in CONFIGS if not isinstance(c[4], str)])       # (sweep MODE 2: block layouts only)
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_bip_apply_matches_numpy(cfg, dt):
    C, T, lo, hi, bt, nwg, er = cfg
    H, N, g = make_backends(C, T, lo, hi, 300 + C, dt, bt, nwg, er)
    rng = np.random.default_rng(2)
    x = np.linalg.qr(rng.standard_normal((3 * (C + T), 3)))[0]      # |x_i|_F <= sqrt(3)
    x[: 3 * C] *= 4.0 if C + T > 40 else 1.0                        # camera and timestep sides of different magnitude
    x[3 * C:] *= 0.25
    zh, zn = H.empty(3 * (C + T), 3), N.empty(3 * (C + T), 3)
    H.bip_scales()
    H.bip_apply(H.from_numpy(x), zh); N.bip_apply(N.from_numpy(x), zn)
    ref = zn.numpy()
    tol = 1e-10 if dt == np.float64 else 2e-6
    zc, zt = zh.cpu().numpy()[: 3 * C], zh.cpu().numpy()[3 * C:]
    assert np.abs(zc - ref[: 3 * C]).max() <= tol * np.abs(ref[: 3 * C]).max()
    assert np.abs(zt - ref[3 * C:]).max() <= tol * np.abs(ref[3 * C:]).max()
    # rows without edges give exact zeros; integer accumulation => bit-identical repeats
    empty = np.repeat(np.diff(N.row_ptr) == 0, 3)
    assert not zt[empty].any()
    z2 = H.empty(3 * (C + T), 3)
    for _ in range(3):
        H.bip_apply(H.from_numpy(x), z2)
        assert torch.equal(zh, z2)
    # symmetry of R~:  <u, R~ v> = <R~ u, v>
    v = np.linalg.qr(rng.standard_normal((3 * (C + T), 3)))[0]
    zv = H.empty(3 * (C + T), 3)
    H.bip_apply(H.from_numpy(v), zv)
    a, b = float((x * zv.cpu().numpy()).sum()), float((zh.cpu().numpy() * v).sum())
    assert abs(a - b) <= (1e-9 if dt == np.float64 else 1e-5) * max(abs(a), abs(b), np.abs(ref).max())
    # the eliminated operator still works on the same graph afterwards (shared fx scales are refreshed by set_duals)
    lam = np.tile(np.eye(3).reshape(1, 9), (T, 1))
    xq = np.linalg.qr(rng.standard_normal((3 * C, 3)))[0]
    ph, pn = H.empty(3 * C, 3), N.empty(3 * C, 3)
    H.set_duals(H.from_numpy(lam)); H.block_op(H.from_numpy(lam), H.from_numpy(xq), ph)
    N.block_op(N.from_numpy(lam), N.from_numpy(xq), pn)
    assert np.abs(ph.cpu().numpy() - pn.numpy()).max() <= tol * np.abs(pn.numpy()).max()


def test_polar_without_det_fix():
    from vican_amd.device import HipBackend  # noqa: F401
    H, N, g = make_backends(5, 40, 1, 3, 7, np.float64)
    rng = np.random.default_rng(3)
    A = rng.standard_normal((500, 3, 3))
    A[::2] *= np.sign(np.linalg.det(A[::2]))[:, None, None] * -1.0          # half the blocks: negative determinant
    Rh, Lh = H.empty(500, 9), H.empty(500, 9)
    H.polar_dual(H.from_numpy(A.reshape(500, 9)), Rh, Lh, 5)
    u, s, vt = np.linalg.svd(A)
    assert np.abs(Rh.cpu().numpy().reshape(500, 3, 3) - u @ vt).max() < 1e-10
    assert np.abs(Lh.cpu().numpy().reshape(500, 3, 3) - (u * s[:, None, :]) @ np.swapaxes(u, 1, 2)).max() < 1e-10
    assert (np.linalg.det(Rh.cpu().numpy().reshape(500, 3, 3)[::2]) < 0).all()
    R1, _ = svd_polar(A, 5)
    assert np.abs(R1 - u @ vt).max() < 1e-14


@pytest.mark.parametrize("name,dt", CASES)
def test_dropin_so3sync_matches_reference(name, dt):
    from vican.bipgo import bipartite_so3sync                       # the shim import path
    src, cons, nr, ff = inputs(name)
    keys, R, evals = golden(name, dt)
    info = {}
    res = bipartite_so3sync(src, constraints=cons, noise_model=nr, edge_filter=ff, maxiter=gc.MAXITER,
                            dtype=np.dtype(dt).type, info=info)
    assert list(res.keys()) == keys
    got = np.stack([res[k] for k in keys])
    assert got.dtype == np.float64
    assert np.abs(got - R).max() < (1e-7 if dt == "float64" else 5e-6)
    ev = np.sort(info["evals"][:, :3], axis=1)
    evr = np.sort(evals, axis=1)[:, :3]
    assert np.abs(ev - evr).max() < (1e-7 if dt == "float64" else 1e-4) * np.abs(evals).max()


def test_so3sync_error_behaviour():
    from vican.bipgo import bipartite_so3sync
    src, cons, nr, ff = inputs("g2_small")
    with pytest.raises(UnboundLocalError):                          # bipgo.py:139 with maxiter = 0
        bipartite_so3sync(src, cons, nr, ff, 0)
    with pytest.raises(ValueError):
        bipartite_so3sync(src, cons, nr, lambda e: False, 2)
    bad = dict(cons); bad.pop(sorted(bad.keys())[-1])
    with pytest.raises(KeyError):                                   # unknown marker id (bipgo.py:41)
        bipartite_so3sync(src, bad, nr, ff, 2)
    one = bipartite_so3sync(src, cons, nr, ff, 1, dtype=np.float32)
    assert next(iter(one.values())).dtype == np.float32            # r stays in the eigs dtype after one iteration


def test_so3sync_interior_regime_matches_the_reference(monkeypatch):
    """Golden g3: the connection Laplacian turns indefinite after the first dual update and the reference goes on with INTERIOR
    eigenvectors (eigs(sigma=-1e-6), bipgo.py:106).  The drop-in notices, starts over and takes every step on the dense Laplacian
    (solver.GeneralRotationSolver._interior_step): the real reference's output, chaotic as the iteration is there; graphs too large
    for a dense matrix are still refused."""
    from vican.bipgo import bipartite_so3sync
    from vican_amd.solver import GeneralRotationSolver
    src3, cons3, nr3, ff3 = inputs("g3_medium")
    keys, R, evals = golden("g3_medium", "float64")
    info = {}
    res = bipartite_so3sync(src3, cons3, nr3, ff3, gc.MAXITER, dtype=np.float64, info=info)
    assert list(res.keys()) == keys and info["interior_from"] == 1
    got = np.stack([res[k] for k in keys])
    assert np.abs(got - R).max() < 1e-6, np.abs(got - R).max()      # (the oracle itself: 1e-7 on the same LAPACK calls; amplification ~1e6)
    ev, evr = np.sort(info["evals"], axis=1), np.sort(evals, axis=1)
    assert ev.shape == evr.shape and np.abs(ev - evr).max() < 1e-8
    monkeypatch.setattr(GeneralRotationSolver, "INTERIOR_MAX_N", 600)
    with pytest.raises(ArithmeticError, match="indefinite"):
        bipartite_so3sync(src3, cons3, nr3, ff3, gc.MAXITER, dtype=np.float64)


def test_so3sync_larger_synthetic_consistent_graph():
    """A noise-free single-marker graph (the regime where the variant is well posed): every relative rotation
    r_c r_t^T of the answer must reproduce the measurement, at a size that exercises the 768-thread sweep."""
    from vican_amd import synth
    from vican_amd.geometry import SE3
    from vican.bipgo import bipartite_so3sync
    scene = synth.make_scene(n_cam=48, n_time=1500, n_marker=1, seed=5)
    flat = synth.make_camera_edges(scene, cpt=24, mpv=1, sigma_r=0.0, sigma_t=0.0, seed=6)
    src = synth.edges_to_dict(flat, SE3)
    cons = synth.constraints_from_scene(scene, SE3)
    res = bipartite_so3sync(src, cons, lambda e: 1.0, lambda e: True, 3, dtype=np.float64)
    worst = 0.0
    for (c, tm), v in list(src.items())[::7]:
        ts = tm.split("_")[0]
        # L r = 0 for consistent data: M_ct r_t = a_ct r_c  =>  R~_e R_m R_0^T r_t = r_c  (bipgo.py:45)
        lhs = v["pose"].R() @ cons[tm.split("_")[1]].R() @ cons[str(min(cons.keys()))].R().T @ res[ts + "_0"]
        worst = max(worst, np.abs(lhs - res[c]).max())
    assert worst < 1e-8, worst


@pytest.mark.parametrize("n,ka", [(16384, 3), (50001, 7), (303000, 48), (600000, 96)])
def test_sliced_tall_gram_for_long_vectors(n, ka):
    """vican_tall_gram with a workspace: row-sliced partial sums folded in slice order (long vectors of the
    non-eliminated solver) against NumPy, bit-identical on repeats; without workspace: the column kernel."""
    H, N, g = make_backends(5, 40, 1, 3, 7, np.float64)
    gen = torch.Generator(device="cuda"); gen.manual_seed(n + ka)
    V = torch.randn(ka * n, generator=gen, dtype=torch.float64, device="cuda")
    R = torch.randn(3 * n, generator=gen, dtype=torch.float64, device="cuda")
    out, out2 = H.zeros(ka * 3), H.zeros(ka * 3)
    H.tall_gram(n, V, n, ka, R, out)
    ref = (V.view(ka, n) @ R.view(3, n).T).cpu().numpy().reshape(-1)
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-10 * np.sqrt(n)
    H.tall_gram(n, V, n, ka, R, out2)
    assert torch.equal(out, out2)
    H._gram_workspace = lambda n_: None                     # no workspace -> one workgroup per column
    H.tall_gram(n, V, n, ka, R, out2)
    assert np.abs(out2.cpu().numpy() - ref).max() < 1e-10 * np.sqrt(n)


def test_general_lanczos_step_long_vectors_matches_numpy():
    C, T = 300, 6000
    H, N, g = make_backends(C, T, 3, 8, 11, np.float64)
    nn = C + T
    n, j, m = 3 * nn, 2, 6
    rng = np.random.default_rng(4)
    Q = np.linalg.qr(rng.standard_normal((n, 3 * (j + 1))))[0]
    V0 = np.zeros((3 * (m + 1), n)); V0[: 3 * (j + 1)] = Q.T
    lam = rng.standard_normal((nn, 3, 3)); lam = lam + np.swapaxes(lam, 1, 2) + 6 * np.eye(3)
    z = rng.standard_normal((n, 3))
    outs = []
    for K in (H, N):
        V = K.from_numpy(V0.reshape(-1).copy())
        R, Hs, G = K.empty(3 * n), K.empty(3 * (m + 1) * 3), K.empty(9)
        Hcol, beta, x = K.zeros(3 * (m + 1) * 3), K.zeros(9), K.empty(n, 3)
        K.lanczos_cam_step(K.from_numpy(lam.reshape(nn, 9)), V, n, j, K.from_numpy(z), R, Hs, G, Hcol, beta, x, 0.0)
        outs.append([t.cpu().numpy() for t in (Hcol, beta, x)])
    for a, b in zip(*outs):
        assert np.abs(a - b).max() < 1e-9 * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5, 7, 96])
def test_random_so3sync_matches_oracle_or_refuses(seed):
    """Random scenes through bipartite_so3sync: single-marker scenes (the consistent regime) must match the oracle;
    multi-marker ones match - also where the dual iterate turns the Laplacian indefinite and the reference's shift-invert picks
    interior eigenvectors (float64: the dense regime of the solver; seed 96: smallest eigenvalue -0.95, fourth 0.36, met after a
    restart whose one-block Krylov space has no fourth Ritz value of its own) - or, in float32 there, raise ArithmeticError (the
    reference's single-precision ARPACK result is not reproducible by itself).  The oracle's randomly started ARPACK is retried
    (util.oracle_attempts); a reference that dies on the scene (NaN eigenvectors -> LinAlgError) ends the comparison."""
    from oracle import bipgo_oracle as orc
    from vican.bipgo import bipartite_so3sync
    from vican_amd import synth
    from vican_amd.geometry import SE3
    rng = np.random.default_rng(5000 + seed)
    n_cam = int(rng.integers(3, 25)); n_time = int(rng.integers(20, 300))
    n_marker = 1 if seed % 3 else int(rng.integers(2, 6))
    scene = synth.make_scene(n_cam=n_cam, n_time=n_time, n_marker=n_marker, seed=seed)
    sig = float(10.0 ** rng.uniform(-4, -2))
    flat = synth.make_camera_edges(scene, cpt=int(min(n_cam, rng.integers(2, 5))), mpv=int(min(n_marker, rng.integers(1, 3))),
                                   sigma_r=sig, sigma_t=sig, seed=seed + 1)
    src = synth.edges_to_dict(flat, SE3)
    cons = synth.constraints_from_scene(scene, SE3)
    dt = np.float32 if seed % 2 else np.float64
    unit, keep = (lambda e: 1.0), (lambda e: True)
    info = {}
    try:
        res = bipartite_so3sync(src, cons, unit, keep, gc.MAXITER, dt, info=info)
    except ArithmeticError:
        assert n_marker > 1 and dt == np.float32
        return
    from util import oracle_attempts

    def check(ref):
        assert list(res) == list(ref)
        # (interior regime: the iteration amplifies rounding ~1e6-fold - the oracle moves by 1e-7 between ARPACK starts itself)
        tol = 1e-5 if info.get("interior_from") is not None else (1e-6 if dt == np.float64 else 2e-5)
        assert max(np.abs(res[k] - ref[k]).max() for k in ref) < tol

    def run():
        try:
            return orc.bipartite_so3sync(src, cons, unit, keep, gc.MAXITER, dt)
        except np.linalg.LinAlgError as exc:             # (NaN eigenvectors from a start vector gone astray: another start)
            raise RuntimeError("reference died: %s" % exc)
    try:
        oracle_attempts(run, check)
    except RuntimeError as exc:
        assert "reference died" in str(exc) and n_marker > 1
