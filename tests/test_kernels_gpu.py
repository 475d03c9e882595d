"""Per-kernel parity on the GPU: every HIP entry point vs the NumPy restatement
(vican_amd/backend_cpu.py) on seeded random inputs, through the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from vican_amd.backend_cpu import NumpyBackend, svd_polar       # noqa: E402
from util import load_golden                            # noqa: E402
from vican_amd import synth                             # noqa: E402


@pytest.fixture(autouse=True, params=["banks", "rows"])
def slot_order(request, monkeypatch):
    """Every test of this file under both orders of the edges inside a chunk (vican_graph_t.slot_order: bank-aware /
    row-major; LocalGraph picks one by the average row length, VICAN_SLOT_ORDER forces it)."""
    monkeypatch.setenv("VICAN_SLOT_ORDER", request.param)
    return request.param


def random_graph(C, T, deg_lo, deg_hi, seed, empty_rows=False):
    rng = np.random.default_rng(seed)
    deg = rng.integers(deg_lo, min(deg_hi, C) + 1, T)
    if empty_rows:
        deg[rng.random(T) < 0.1] = 0
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    col = np.concatenate([np.sort(rng.choice(C, d, replace=False)) for d in deg] + [np.zeros(0, int)]).astype(np.int32)
    E = len(col)
    a = rng.uniform(0.5, 2.0, E)
    blk = (synth.random_rotations(rng, E) * a[:, None, None] + 0.05 * rng.standard_normal((E, 3, 3))).reshape(E, 9)
    w = rng.uniform(0.5, 2.0, E)
    u = rng.standard_normal((E, 3)); v = rng.standard_normal((E, 3))
    return rp, col, blk, a, w, u, v


def make_backends(C, T, deg_lo, deg_hi, seed, dt, block_threads=None, n_wg=None, empty_rows=False):
    from vican_amd.device import HipBackend, LocalGraph
    rp, col, blk, a, w, u, v = random_graph(C, T, deg_lo, deg_hi, seed, empty_rows)
    tdt = torch.float32 if dt == np.float32 else torch.float64
    dev = torch.device("cuda:0")
    # block_threads: a workgroup size = block layout; "wave" / "wave4|8|12" = one wavefront per chunk (vican_wsweep.hip)
    kw = dict(block_threads=block_threads, layout="block" if block_threads else None)
    if isinstance(block_threads, str):
        kw = dict(layout="wave", wg_waves=int(block_threads[4:]) if len(block_threads) > 4 else None)
    g = LocalGraph(C, torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev),
                   torch.from_numpy(blk).to(dev, tdt), torch.from_numpy(a).to(dev, tdt),
                   torch.from_numpy(w).to(dev), torch.from_numpy(u).to(dev), torch.from_numpy(v).to(dev), n_wg=n_wg, keep_csr=True, **kw)
    return HipBackend(g), NumpyBackend(C, rp, col, blk, a, w, u, v, storage=dt), g


CONFIGS = [  # C, T, deg_lo, deg_hi, block_threads, n_wg, empty_rows
    (5, 40, 1, 3, 256, None, False),
    (37, 300, 1, 12, 256, None, True),
    (64, 500, 20, 64, 256, 7, False),
    (200, 120, 100, 200, 1024, None, False),
    (300, 200, 50, 300, 512, 9, False),
    (400, 300, 100, 400, 768, 6, False),
    (700, 64, 300, 700, 1024, 5, False),
    # wave layout: rows of at most 256 (f32) / 128 (f64) edges
    (5, 40, 1, 3, "wave", None, False),
    (37, 300, 1, 12, "wave4", None, True),
    (64, 500, 20, 64, "wave8", 7, False),
    (300, 900, 60, 128, "wave12", 9, False),
    (1000, 200, 100, 128, "wave12", None, False),
    (100, 4000, 2, 9, "wave12", 5, False),
    # 1024 cameras, short rows: 63 rows per chunk (camera 1023 of a row 63 would read as the padding word of the 2-byte index)
    (1024, 700, 1, 4, "wave4", None, False),
]


@pytest.mark.parametrize("cfg", CONFIGS)
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_block_op_dual_update_and_init(cfg, dt):
    C, T, lo, hi, bt, nwg, er = cfg
    H, N, g = make_backends(C, T, lo, hi, 100 + C, dt, bt, nwg, er)
    rng = np.random.default_rng(1)
    x = np.linalg.qr(rng.standard_normal((3 * C, 3)))[0]            # |x_c|_F <= sqrt(3): the sweep's precondition
    # initial duals
    lamT_h, cd_h = H.empty(T, 9), H.empty(C)
    lamT_n, cd_n = N.empty(T, 9), N.empty(C)
    H.init_duals(lamT_h, cd_h); N.init_duals(lamT_n, cd_n)
    nz = np.diff(N.row_ptr) > 0
    assert np.allclose(lamT_h.cpu().numpy()[nz], lamT_n.numpy()[nz], rtol=1e-13, atol=0)
    assert np.allclose(cd_h.cpu().numpy(), cd_n.numpy(), rtol=1e-13)
    # random SPD-ish duals for rows with edges
    lam = rng.standard_normal((T, 3, 3)); lam = lam @ np.swapaxes(lam, 1, 2) + np.eye(3)
    lam_h, lam_n = H.from_numpy(lam.reshape(T, 9)), N.from_numpy(lam.reshape(T, 9))
    zh, zn = H.empty(3 * C, 3), N.empty(3 * C, 3)
    H.set_duals(lam_h)                      # duals not produced by the library: refresh the fixed-point bound
    H.block_op(lam_h, H.from_numpy(x), zh); N.block_op(lam_n, N.from_numpy(x), zn)
    ref = zn.numpy()
    # f64 blocks: f64 products + exact fixed-point sums; f32 blocks: f32 products (x rounded to f32)
    # (fixed-point resolution is 2^-47 of the contribution BOUND |M|max*omega*|x|max, not of each value)
    assert np.abs(zh.cpu().numpy() - ref).max() <= (1e-10 if dt == np.float64 else 2e-6) * np.abs(ref).max()
    # integer accumulation => bit-identical on a repeat launch
    z2 = H.empty(3 * C, 3)
    H.block_op(lam_h, H.from_numpy(x), z2)
    assert torch.equal(zh, z2)
    # dual update from stacked rotations
    rc = synth.random_rotations(rng, C).reshape(3 * C, 3)
    Rt_h, L_h = H.zeros(T, 9), H.zeros(T, 9)
    Rt_n, L_n = N.zeros(T, 9), N.zeros(T, 9)
    H.dual_update(H.from_numpy(rc), Rt_h, L_h); N.dual_update(N.from_numpy(rc), Rt_n, L_n)
    well = nz & (np.diff(N.row_ptr) >= 2)            # a single noisy block may be ill-conditioned; keep generic rows
    tol = 1e-9 if dt == np.float64 else 2e-3       # fixed-point y sums / f32 products; S^-1 amplifies by cond(Z_t)
    assert np.abs(Rt_h.cpu().numpy()[well] - Rt_n.numpy()[well]).max() < tol
    rel = np.abs(L_h.cpu().numpy()[well] - L_n.numpy()[well]).max() / np.abs(L_n.numpy()[well]).max()
    assert rel < tol


def test_polar_dual_against_reference_project_SO3():
    """G6 golden: outputs of the REAL reference's project_SO3, incl. reflections,
    near-rank-2 and repeated singular values."""
    from vican_amd.device import HipBackend
    H, _, _ = make_backends(5, 8, 1, 3, 7, np.float64)
    g6 = load_golden("g6_polar")
    x = g6["x"]
    n = len(x)
    R, L1, L2 = H.empty(n, 9), H.empty(n, 9), H.empty(n, 9)
    xin = H.from_numpy(x.reshape(n, 9))
    H.polar_dual(xin, R, L1, 1); H.polar_dual(xin, None, L2, 2)
    Rh = R.cpu().numpy().reshape(n, 3, 3)
    assert np.abs(Rh - g6["project_SO3"]).max() < 1e-6          # near-rank-2 blocks: sigma3 ~ 1e-9
    generic = np.r_[0:96, 128:n]
    assert np.abs(Rh[generic] - g6["project_SO3"][generic]).max() < 1e-12
    _, l1 = svd_polar(x, 1); _, l2 = svd_polar(x, 2)
    assert np.abs(L1.cpu().numpy().reshape(n, 3, 3) - l1).max() < 1e-11
    g2 = np.r_[0:96, 128:n]
    rel = np.abs(L2.cpu().numpy().reshape(n, 3, 3)[g2] - l2[g2]).max() / np.abs(l2[g2]).max()
    assert rel < 1e-10
    assert np.allclose(np.linalg.det(Rh), 1.0, atol=1e-12)


def test_gauge_project_and_lanczos_helpers():
    C = 50
    H, N, _ = make_backends(C, 100, 2, 6, 3, np.float64)
    rng = np.random.default_rng(5)
    n, m = 3 * C, 6
    X = rng.standard_normal((n, 3))
    oh, on = H.empty(n, 3), N.empty(n, 3)
    H.gauge_project(H.from_numpy(X), oh); N.gauge_project(N.from_numpy(X), on)
    assert np.abs(oh.cpu().numpy() - on.numpy()).max() < 1e-10
    assert np.abs(oh.cpu().numpy()[:3] - np.eye(3)).max() < 1e-12
    # basis helpers
    Vn = rng.standard_normal((3 * m, n))
    Vh, Vc = H.from_numpy(Vn.reshape(-1).copy()), N.from_numpy(Vn.reshape(-1).copy())
    R0 = rng.standard_normal((3, n))
    Rh, Rn = H.from_numpy(R0.reshape(-1).copy()), N.from_numpy(R0.reshape(-1).copy())
    ka = 3 * m - 3
    Hh, Hn = H.empty(3 * m * 3), N.empty(3 * m * 3)
    H.tall_gram(n, Vh, n, ka, Rh, Hh); N.tall_gram(n, Vc, n, ka, Rn, Hn)
    assert np.abs(Hh.cpu().numpy()[: ka * 3] - Hn.numpy()[: ka * 3]).max() < 1e-11
    Ho_h, Ho_n = H.zeros(3 * m * 3), N.zeros(3 * m * 3)
    for acc in (0, 1):
        H.tall_update(n, Vh, n, ka, Hh, Rh, Ho_h, acc); N.tall_update(n, Vc, n, ka, Hn, Rn, Ho_n, acc)
    assert np.abs(Rh.cpu().numpy() - Rn.numpy()).max() < 1e-9
    assert np.abs(Ho_h.cpu().numpy() - Ho_n.numpy()).max() < 1e-10
    Gh, Gn = H.empty(9), N.empty(9)
    H.tall_gram(n, Rh, n, 3, Rh, Gh); N.tall_gram(n, Rn, n, 3, Rn, Gn)
    bh, bn, xh, xn = H.empty(9), N.empty(9), H.empty(n, 3), N.empty(n, 3)
    H.chol_qr3(n, Rh, Gh, Vh, n, ka, bh, xh, 0.0); N.chol_qr3(n, Rn, Gn, Vc, n, ka, bn, xn, 0.0)
    assert np.abs(bh.cpu().numpy() - bn.numpy()).max() < 1e-9 * np.abs(bn.numpy()).max()
    q = xh.cpu().numpy()
    assert np.abs(q - xn.numpy()).max() < 1e-10
    assert np.abs(q.T @ q - np.eye(3)).max() < 1e-12
    assert np.abs(Vh.cpu().numpy().reshape(3 * m, n)[ka:ka + 3] - q.T).max() == 0.0
    Y = rng.standard_normal((3 * m, 3))
    Xh, Xn = H.empty(n, 3), N.empty(n, 3)
    H.tall_combine(n, Vh, n, 3 * m, H.from_numpy(Y), Xh); N.tall_combine(n, Vc, n, 3 * m, N.from_numpy(Y), Xn)
    assert np.abs(Xh.cpu().numpy() - Xn.numpy()).max() < 1e-11
    lam = rng.standard_normal((C, 9)); z = rng.standard_normal((n, 3))
    ah, an = H.empty(3 * n), N.empty(3 * n)
    H.lap_apply(H.from_numpy(lam), Vh, n, 3, H.from_numpy(z), ah); N.lap_apply(N.from_numpy(lam), Vc, n, 3, N.from_numpy(z), an)
    assert np.abs(ah.cpu().numpy() - an.numpy()).max() < 1e-11
    H.rows_to_cols(n, H.from_numpy(X), Vh, n, 0); N.rows_to_cols(n, N.from_numpy(X), Vc, n, 0)
    assert np.array_equal(Vh.cpu().numpy()[: 3 * n], Vc.numpy()[: 3 * n])
    # vanished pivot -> zero column + zero beta (breakdown signalling)
    H.chol_qr3(n, Rh, H.from_numpy(np.diag([1.0, 1e-40, 1.0]).reshape(-1)), Vh, n, 0, bh, xh, 1e-20)
    b = bh.cpu().numpy().reshape(3, 3)
    assert b[1, 1] == 0.0 and np.all(xh.cpu().numpy()[:, 1] == 0.0)


@pytest.mark.parametrize("cfg", CONFIGS[:6] + CONFIGS[8:11])
def test_translation_kernels_and_cg(cfg):
    """rhs / degrees / every CG kernel step by step against the NumPy state machine."""
    from vican_amd.solver import Comm, TranslationSolver
    C, T, lo, hi, bt, nwg, er = cfg
    H, N, g = make_backends(C, T, lo, hi, 200 + C, np.float64, bt, nwg, False)
    rng = np.random.default_rng(2)
    rc = synth.random_rotations(rng, C).reshape(3 * C, 3)
    rt = synth.random_rotations(rng, T).reshape(T, 9)
    outs = []
    for K in (H, N):
        ts = TranslationSolver(K, Comm(), rtol=1e-8, poll_every=4)
        ts.setup(K.from_numpy(rc), K.from_numpy(rt))
        x_c, x_t = ts.solve(3 * (C + T))
        outs.append((ts.deg_t.cpu().numpy(), ts.deg_c.cpu().numpy(), ts.b_t.cpu().numpy(), ts.b_c.cpu().numpy(),
                     x_c.cpu().numpy(), x_t.cpu().numpy(), ts.info))
    (dt_h, dc_h, bt_h, bc_h, xc_h, xt_h, ih), (dt_n, dc_n, bt_n, bc_n, xc_n, xt_n, inn) = outs
    assert np.allclose(dt_h, dt_n, rtol=1e-13) and np.allclose(dc_h, dc_n, rtol=1e-13)
    assert np.abs(bt_h - bt_n).max() < 1e-11 and np.abs(bc_h - bc_n).max() < 1e-10
    assert ih["converged"] and inn["converged"]
    assert abs(ih["cg_iters"] - inn["cg_iters"]) <= max(1, inn["cg_iters"] // 20)     # rtol 1e-8: ~100 iterations
    scale = max(np.abs(xt_n).max(), 1.0)
    assert np.abs(xc_h - xc_n).max() < 1e-6 * scale and np.abs(xt_h - xt_n).max() < 1e-6 * scale
    # translations from x0 = 0 stay in the range of the Laplacian: node sum is zero
    assert np.abs(xc_h.sum(0) + xt_h.sum(0)).max() < 1e-8 * scale * (C + T)


@pytest.mark.parametrize("cfg", [CONFIGS[1], CONFIGS[4], CONFIGS[8], CONFIGS[10], CONFIGS[11], CONFIGS[12], (1000, 3000, 250, 250, "wave12", None, False)])
@pytest.mark.parametrize("rtol", [1e-5, 1e-9])
def test_fused_cg_iteration(cfg, rtol):
    """vican_cg_iter_fused (begin, sweep, a fold that forms p_t.q_t over fixed slices, the step) against the launch sequence
    vican_cg_iter_local / vican_cg_iter_finish.
      * run under UNEVEN load (a filler kernel occupies part of the chip on another stream while the iterations run) the
        iterates must be bit-identical from run to run - the sweep's own p.q partial (ticket order) was not;
      * against the sequence: the same recurrence, the timestep part of p.q grouped differently (and, in the sequence, in the
        order the sweep's tickets happened to fall): same iterates to rounding, same iteration count on these well-conditioned
        systems;
      * iteration budget as scipy's range(maxiter)."""
    from test_coop_barriers_gpu import occupy
    from vican_amd.solver import Comm, TranslationSolver
    C, T, lo, hi, bt, nwg, er = cfg
    if isinstance(bt, str) and hi > 128 and bt.startswith("wave"):
        dtp = np.float32                                         # (rows of 250 edges fit a wave chunk only with float32 blocks)
    else:
        dtp = np.float64
    H, N, g = make_backends(C, T, lo, hi, 300 + C, dtp, bt, nwg, False)
    H._cgres_ok = False                                          # (capture-sized graphs: the launch paths, not the resident kernel)
    rng = np.random.default_rng(3)
    rc = synth.random_rotations(rng, C).reshape(3 * C, 3)
    rt = synth.random_rotations(rng, T).reshape(T, 9)
    names = ("x_c", "x_t", "r_c", "r_t", "p_c", "p_t", "q_t")
    side = torch.cuda.Stream()

    def run_fused(n_iter, load):
        ts = TranslationSolver(H, Comm.single(), rtol=rtol)
        ts.setup(H.from_numpy(rc), H.from_numpy(rt))
        H.cg_init(ts.b_c, ts.b_t, ts.x_c, ts.x_t, ts.r_c, ts.r_t, ts.p_c, ts.p_t, ts.st)
        torch.cuda.synchronize()
        states = []
        for k in range(n_iter):
            if load and k % 2 == 0:
                occupy(H.lib, 64 + 37 * (k % 5), 40 + 30 * (k % 3), side)      # part of the chip busy for 40-100 us
            H.cg_iter_fused(ts.deg_t, ts.deg_c, ts.r_c, ts.p_c, ts.x_c, ts.r_t, ts.p_t, ts.q_t, ts.x_t, ts.qcpq, rtol, ts.st, first=(k == 0))
            if k in (0, 3, n_iter - 1):
                torch.cuda.synchronize()
                states.append(([getattr(ts, nm).clone() for nm in names], ts._state()))
        torch.cuda.synchronize()
        return states
    ref = run_fused(10, False)
    for rep in range(4):
        got = run_fused(10, rep > 0)
        for (va, sa), (vb, sb) in zip(ref, got):
            for nm, a, b in zip(names, va, vb):
                assert torch.equal(a, b), (rep, nm)
            assert all(sa[k] == sb[k] or (sa[k] != sa[k] and sb[k] != sb[k]) for k in sa), (rep, sa, sb)
    # against the launch sequence, iteration by iteration (its head of the NEXT iteration applied: the fused step has run it)
    tf, tq = TranslationSolver(H, Comm.single(), rtol=rtol), TranslationSolver(H, Comm.single(), rtol=rtol)
    for ts in (tf, tq):
        ts.setup(H.from_numpy(rc), H.from_numpy(rt))
        H.cg_init(ts.b_c, ts.b_t, ts.x_c, ts.x_t, ts.r_c, ts.r_t, ts.p_c, ts.p_t, ts.st)
    H.cg_begin(tq.r_c, tq.p_c, rtol, tq.st, 0)
    for k in range(6):
        H.cg_iter_fused(tf.deg_t, tf.deg_c, tf.r_c, tf.p_c, tf.x_c, tf.r_t, tf.p_t, tf.q_t, tf.x_t, tf.qcpq, rtol, tf.st, first=(k == 0))
        H.cg_sweep(tq.deg_t, tq.p_c, tq.r_t, tq.p_t, tq.q_t, tq.qcpq, tq.st)
        n_part = H.cg_iter_finish(tq.deg_c, tq.qcpq, tq.p_c, tq.x_c, tq.r_c, tq.p_t, tq.q_t, tq.x_t, tq.r_t, tq.st)
        H.cg_begin(tq.r_c, tq.p_c, rtol, tq.st, n_part)
        torch.cuda.synchronize()
        sf, sq = tf._state(), tq._state()
        # (the fused call has not run the next head yet: p_c and the head's scalars are compared a call later)
        for nm in [x for x in names if x != "p_c"]:
            a, b = getattr(tf, nm), getattr(tq, nm)
            assert float((a - b).abs().max()) <= 1e-11 * max(float(b.abs().max()), 1e-300), (k, nm)
        for key in ("alpha", "pq", "rr_cam"):
            assert abs(sf[key] - sq[key]) <= 1e-11 * abs(sq[key]), (k, key, sf[key], sq[key])
        if sq["done"]:
            break
    # whole solves: the polled driver on either path
    outs = []
    for fused in (True, False):
        ts = TranslationSolver(H, Comm.single(), rtol=rtol, poll_every=4)
        ts.use_fused = fused                                     # (False: the launch sequence the camera-tiled backends take)
        ts.setup(H.from_numpy(rc), H.from_numpy(rt))
        x_c, x_t = ts.solve(3 * (C + T))
        outs.append((x_c.clone(), x_t.clone(), dict(ts.info)))
    assert outs[0][2]["converged"] and abs(outs[0][2]["cg_iters"] - outs[1][2]["cg_iters"]) <= max(1, outs[1][2]["cg_iters"] // 20)
    if outs[0][2]["cg_iters"] == outs[1][2]["cg_iters"]:
        scale = max(float(outs[1][1].abs().max()), 1.0)
        assert float((outs[0][0] - outs[1][0]).abs().max()) < 1e3 * rtol * scale and float((outs[0][1] - outs[1][1]).abs().max()) < 1e3 * rtol * scale
    # iteration budget (scipy: range(maxiter), no test behind the last update)
    ts = TranslationSolver(H, Comm.single(), rtol=1e-14)
    ts.setup(H.from_numpy(rc), H.from_numpy(rt))
    ts.solve(3 * (C + T), maxiter=3)
    assert ts.info["cg_iters"] == 3 and not ts.info["converged"]


@pytest.mark.parametrize("cfg", [CONFIGS[7], CONFIGS[8], CONFIGS[9], CONFIGS[10], CONFIGS[12], (340, 10000, 2, 6, "wave", None, False),
                                 (200, 30000, 2, 6, "wave", None, False)])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_cg_resident_matches_multikernel(cfg, dt):
    """vican_cg_resident (the whole CG as one cooperative launch, vican_cgres.hip) against the multi-kernel path on the same
    device: same iteration count, iterates equal to rounding (the floating-point partial sums are grouped differently),
    bit-identical on repeats; the Jacobi-scaled tight solver through both; an iteration budget is honoured exactly."""
    from vican_amd.solver import Comm, TightTranslationSolver, TranslationSolver
    C, T, lo, hi, bt, nwg, er = cfg
    H, N, g = make_backends(C, T, lo, hi, 200 + C, dt, bt, nwg, er)
    assert H.cg_resident_ok == (g.n_wg <= 128)
    H._cgres_ok = True                                       # (also above the size where the solver prefers it)
    rng = np.random.default_rng(2)
    rc = synth.random_rotations(rng, C).reshape(3 * C, 3)
    rt = synth.random_rotations(rng, T).reshape(T, 9)
    for cls, rtol in ((TranslationSolver, 1e-8), (TranslationSolver, 1e-5), (TightTranslationSolver, 1e-10)):
        outs = []
        for resident in (True, False, True):
            H._cgres_ok = resident
            ts = cls(H, Comm(), rtol=rtol)
            ts.small_graph = True
            ts.setup(H.from_numpy(rc), H.from_numpy(rt))
            x_c, x_t = ts.solve(3 * (C + T))
            assert bool(ts.info.get("resident", False)) == resident and ts.info["converged"]
            outs.append((x_c.clone(), x_t.clone(), ts.info["cg_iters"], ts.info["relres"]))
        H._cgres_ok = True
        scale = max(float(outs[1][1].abs().max()), 1.0)
        # (the two paths round their floating-point partial sums differently - the multi-kernel sweep pre-sums a lane's same-row
        #  terms in f64 - and the stopping test is a threshold: a few per cent of the iteration count, as against the NumPy stand-in)
        assert abs(outs[0][2] - outs[1][2]) <= max(1, outs[1][2] // 20), (outs[0][2], outs[1][2])
        if outs[0][2] == outs[1][2]:
            # (both iterates satisfy |r| < rtol |b|; rounding differences of the recurrences are amplified towards that level)
            tol = 1e3 * rtol * scale
            assert float((outs[0][0] - outs[1][0]).abs().max()) < tol and float((outs[0][1] - outs[1][1]).abs().max()) < tol
        assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1]) and outs[0][2] == outs[2][2]
    # iteration budget: exactly maxiter updates, not converged (scipy: info = maxiter)
    ts = TranslationSolver(H, Comm(), rtol=1e-14)
    ts.small_graph = True
    ts.setup(H.from_numpy(rc), H.from_numpy(rt))
    ts.solve(3 * (C + T), maxiter=3)
    assert ts.info["cg_iters"] == 3 and not ts.info["converged"] and ts.info["resident"]
    x3 = ts.x_t.cpu().numpy().copy()
    tn = TranslationSolver(N, Comm(), rtol=1e-14)
    tn.setup(N.from_numpy(rc), N.from_numpy(rt))
    # the NumPy state machine run for the same three updates
    N.cg_init(tn.b_c, tn.b_t, tn.x_c, tn.x_t, tn.r_c, tn.r_t, tn.p_c, tn.p_t, tn.st)
    n_part = 0
    for _ in range(3):
        N.cg_iter_local(tn.deg_t, tn.r_c, tn.p_c, tn.r_t, tn.p_t, tn.q_t, tn.qcpq, 1e-14, tn.st, n_part)
        n_part = N.cg_iter_finish(tn.deg_c, tn.qcpq, tn.p_c, tn.x_c, tn.r_c, tn.p_t, tn.q_t, tn.x_t, tn.r_t, tn.st)
    assert np.abs(x3 - tn.x_t.numpy()).max() < 1e-9 * max(np.abs(x3).max(), 1e-30)


@pytest.mark.parametrize("cfg", [CONFIGS[1], CONFIGS[2], CONFIGS[4], CONFIGS[9]])
def test_lsqr_kernels(cfg):
    """LSQR translation solve: HIP kernels vs the NumPy restatement, same host driver."""
    from vican_amd.solver import Comm, LsqrTranslationSolver
    C, T, lo, hi, bt, nwg, er = cfg
    H, N, g = make_backends(C, T, lo, hi, 300 + C, np.float64, bt, nwg, False)
    rng = np.random.default_rng(3)
    rc = synth.random_rotations(rng, C).reshape(3 * C, 3)
    rt = synth.random_rotations(rng, T).reshape(T, 9)
    outs = []
    for K in (H, N):
        ls = LsqrTranslationSolver(K, Comm(), atol=1e-10, btol=1e-10)
        x_c, x_t = ls.solve(K.from_numpy(rc), K.from_numpy(rt), 3 * (C + T))
        outs.append((x_c.cpu().numpy(), x_t.cpu().numpy(), ls.info))
    (xc_h, xt_h, ih), (xc_n, xt_n, inn) = outs
    assert ih["istop"] == inn["istop"] and abs(ih["lsqr_iters"] - inn["lsqr_iters"]) <= max(1, inn["lsqr_iters"] // 20)
    scale = max(np.abs(xt_n).max(), 1.0)
    assert np.abs(xc_h - xc_n).max() < 1e-6 * scale and np.abs(xt_h - xt_n).max() < 1e-6 * scale
    # the HIP run above kept every scalar on the device and made one fused pass over the edges per iteration (vican_lsqr_step);
    # the round-2 path - Golub-Kahan scalars on the host, two passes - must stop at the same iteration with the same answer
    assert ih.get("device_scalars")
    H.lsqr_host_scalars = True
    ls = LsqrTranslationSolver(H, Comm(), atol=1e-10, btol=1e-10)
    x_c, x_t = ls.solve(H.from_numpy(rc), H.from_numpy(rt), 3 * (C + T))
    assert not ls.info.get("device_scalars") and ls.info["istop"] == ih["istop"]
    assert abs(ls.info["lsqr_iters"] - ih["lsqr_iters"]) <= max(1, ih["lsqr_iters"] // 20)      # (tolerance 1e-10: ~90 iterations)
    assert np.abs(x_c.cpu().numpy() - xc_h).max() < 1e-7 * scale and np.abs(x_t.cpu().numpy() - xt_h).max() < 1e-7 * scale


@pytest.mark.parametrize("C,j", [(5, 0), (60, 3), (333, 7), (1000, 5), (1024, 20)])
def test_lanczos_cam_step_variants_agree(C, j):
    """Camera-side Lanczos step: the cooperative single-kernel variant (grid barriers), the launch sequence /
    single-workgroup kernel behind vican_lanczos_cam_step and the NumPy restatement give the same block."""
    H, N, _ = make_backends(C, 40, 1, min(C, 6), 9, np.float64)
    rng = np.random.default_rng(11)
    n, m = 3 * C, 24
    ka = min(3 * (j + 1), n - 3)
    j = ka // 3 - 1
    Q, _ = np.linalg.qr(rng.standard_normal((n, 3 * (j + 1))))
    Vn = np.zeros((3 * (m + 1), n)); Vn[: 3 * (j + 1)] = Q.T
    lam = rng.standard_normal((C, 3, 3)); lam = (lam @ np.swapaxes(lam, 1, 2) + np.eye(3)).reshape(C, 9)
    z = rng.standard_normal((n, 3))
    outs = []
    for K, coop in ((H, True), (H, True), (H, False), (N, None)):          # the cooperative kernel twice: re-armed barrier
        if coop is not None:
            K.coop_cam_step = coop
        V = K.from_numpy(Vn.reshape(-1).copy())
        R, Hs, G, Hcol, beta, x = K.zeros(3 * n), K.zeros(3 * (m + 1) * 3), K.zeros(9), K.zeros(3 * (m + 1) * 3), K.zeros(9), K.zeros(n, 3)
        K.lanczos_cam_step(K.from_numpy(lam), V, n, j, K.from_numpy(z), R, Hs, G, Hcol, beta, x, 0.0)
        outs.append([t.cpu().numpy() for t in (V, Hcol, beta, x)])
    H.coop_cam_step = True
    for o in outs[:3]:
        for a, b in zip(o, outs[3]):
            assert np.abs(a - b).max() < 1e-10 * max(1.0, np.abs(b).max())
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)                                 # deterministic
    q = outs[0][3]
    assert np.abs(q.T @ q - np.eye(3)).max() < 1e-12 and np.abs(Q.T @ q).max() < 1e-12


def test_cooperative_step_repeats_bit_identically():
    """The cooperative kernel's grid barrier and its cross-workgroup partials use relaxed agent-scope atomics (a release
    fence would write back the XCD's whole L2): 3000 back-to-back repetitions on the same inputs
    must reproduce the first result bit for bit (a workgroup reading stale partials would not), with other kernels
    dirtying the caches in between, and a barrier counter left non-zero by an aborted launch is re-armed by the next
    eigen-solve (lanczos_seed zeroes it)."""
    C, j = 1000, 6
    H, N, _ = make_backends(C, 40, 1, 6, 9, np.float64)
    rng = np.random.default_rng(12)
    n, m = 3 * C, 24
    Q, _ = np.linalg.qr(rng.standard_normal((n, 3 * (j + 1))))
    Vn = np.zeros((3 * (m + 1), n)); Vn[: 3 * (j + 1)] = Q.T
    lam = rng.standard_normal((C, 3, 3)); lam = (lam @ np.swapaxes(lam, 1, 2) + np.eye(3)).reshape(C, 9)
    lamd, zd, V0 = H.from_numpy(lam), H.from_numpy(rng.standard_normal((n, 3))), H.from_numpy(Vn.reshape(-1).copy())
    V = V0.clone()
    R, Hs, G, Hcol, beta, x = H.zeros(3 * n), H.zeros(3 * (m + 1) * 3), H.zeros(9), H.zeros(3 * (m + 1) * 3), H.zeros(9), H.zeros(n, 3)
    H.coop_cam_step = True
    H.lanczos_cam_step(lamd, V, n, j, zd, R, Hs, G, Hcol, beta, x, 0.0)
    ref = [t.clone() for t in (V, Hcol, beta, x)]
    junk = torch.empty(1 << 22, dtype=torch.float64, device="cuda")
    bad = 0
    for rep in range(3000):
        V.copy_(V0)
        if rep % 7 == 0:
            junk.normal_()                                   # other traffic through the caches
        H.lanczos_cam_step(lamd, V, n, j, zd, R, Hs, G, Hcol, beta, x, 0.0)
        if rep % 50 == 49 or rep < 20:
            bad += sum(int(not torch.equal(a, b)) for a, b in zip((V, Hcol, beta, x), ref))
    assert bad == 0
    # an aborted launch leaves the counter non-zero: the next eigen-solve's seed re-arms it
    H._coop_sync.fill_(5)
    x0 = H.from_numpy(rng.standard_normal((n, 3)))
    H.lanczos_seed(x0, V, n, beta, x)
    assert int(H._coop_sync.abs().sum()) == 0


def test_wave_sweeps_repeat_bit_identically():
    """The sweeps' sums are fixed-point, so their results do not depend on which workgroup took which chunk - as long as
    every chunk is taken exactly once.  The range scheduler of wave_sweep_kernel once published "queue exhausted" BEFORE the
    last range (two LDS stores in the wrong order): a wavefront polling between them left with its ticket inside that range
    and the chunk's rows were missing from the sums, in about one sweep of 10^4 on the stress graph (a Lanczos solve then
    needed extra steps; results still converged).  Tens of thousands of back-to-back sweeps on the same inputs must
    reproduce the first result bit for bit, for the operator sweep (MODE 0 + fold) and the fused dual update (MODE 3)."""
    from vican_amd.device import HipBackend, LocalGraph
    dev = torch.device("cuda:0")
    C, T, cpt = 1000, 20000, 250
    gr = synth.make_merged_graph_torch(C, T, cpt, dev, torch.float32, seed=5)
    K = HipBackend(LocalGraph(C, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"]))
    del gr
    assert K.g.layout == "wave"
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    rc = torch.randn(3 * C, 3, dtype=torch.float64, device=dev, generator=gen)
    Rt, lamT, z = K.empty(T, 9), K.empty(T, 9), K.empty(3 * C, 3)
    K.dual_update_op(rc, Rt, lamT, z)
    x, z0 = torch.randn(3 * C, 3, dtype=torch.float64, device=dev, generator=gen), K.empty(3 * C, 3)
    lam_keep = lamT.clone()

    def repeat(fn, outs, n):
        fn(); torch.cuda.synchronize()
        ref = [o.clone() for o in outs]
        cnt = torch.zeros((), dtype=torch.int64, device=dev)
        for _ in range(n):
            fn()
            for o, r in zip(outs, ref):
                cnt += (o != r).any()
        return int(cnt)

    assert repeat(lambda: K.block_op(lam_keep, x, z0), [z0], 30000) == 0
    assert repeat(lambda: K.dual_update_op(rc, Rt, lamT, z), [Rt, lamT, z], 15000) == 0


def _ritz_inputs(steps, m, seed, dead_at=None, gap=True):
    """HB rows as vican_lanczos_cam_step writes them, for a random symmetric projected matrix."""
    rng = np.random.default_rng(seed)
    ka = 3 * steps
    Q = np.linalg.qr(rng.standard_normal((ka, ka)))[0]
    ev = np.sort(rng.uniform(1.0, 50.0, ka))
    if gap and ka > 3:
        ev[:3] = [0.011, 0.013, 0.02]                    # the cluster the solver is after
    Tm = (Q * ev) @ Q.T
    hw = 3 * (m + 1) * 3
    HB = np.zeros((m, hw + 9))
    for j in range(steps):
        kj = 3 * (j + 1)
        HB[j, : kj * 3] = Tm[:kj, 3 * j:3 * j + 3].reshape(-1)
        HB[j, hw:] = np.triu(rng.standard_normal((3, 3)) * 1e-3 + np.eye(3) * 1e-2).reshape(-1)
    if dead_at is not None:
        HB[dead_at, hw + 8] = 0.0
    return HB, hw


@pytest.mark.parametrize("steps,dead_at", [(1, None), (2, None), (5, None), (6, None), (7, 4), (9, None), (16, None), (32, None), (3, 2)])
def test_ritz_matches_lapack_and_gate_semantics(steps, dead_at):
    H, N, g = make_backends(5, 40, 1, 3, 7, np.float64)
    m = 32
    HB, hw = _ritz_inputs(steps, m, 50 + steps, dead_at)
    for flags, eig_tol, floor_tol, level, prev in [(1, 1e-10, 1e-7, -1.0, 0.0), (0, 1e-10, 1.0, -1.0, 1e-3), (2, 1e-10, 1e-7, -1.0, 0.0),
                                                   (1, 1.0, 1e-7, -1.0, 0.0), (1, 1e-10, 1e-7, 10.0, 0.0)]:
        HBd = H.from_numpy(HB)
        Yd, std, gd = H.zeros(3 * (m + 1), 3), H.zeros(16), H.zeros(1, dtype=torch.int32)
        Yn, stn, gn = N.zeros(3 * (m + 1), 3), N.zeros(16), N.zeros(1, dtype=torch.int32)
        std[12] = prev; stn[12] = prev
        Yd.fill_(7.0)                                        # rows beyond the basis must be overwritten with zeros
        H.ritz(HBd, hw, steps, flags, eig_tol, floor_tol, level, Yd, std, gd)
        N.ritz(torch.from_numpy(HB), hw, steps, flags, eig_tol, floor_tol, level, Yn, stn, gn)
        sd, sn = std.cpu().numpy(), stn.numpy()
        assert int(gd.item()) == int(gn[0]), (flags, sd, sn)
        np.testing.assert_array_equal(sd[2:7], sn[2:7])      # stop, converged, floor_hit, eff, breakdown
        np.testing.assert_allclose(sd[7:12], sn[7:12], rtol=1e-12, atol=1e-13, equal_nan=True)
        np.testing.assert_allclose(sd[[13, 15]], sn[[13, 15]], rtol=1e-12, atol=1e-13, equal_nan=True)   # 5th and 4th smallest
        np.testing.assert_allclose(sd[[0, 1, 14]], sn[[0, 1, 14]], rtol=1e-7)
        assert sd[12] == sd[0]
        eff = int(sn[5])
        yd, yn = Yd.cpu().numpy(), Yn.numpy()
        assert np.all(yd[3 * eff:3 * steps] == 0.0)
        k = min(3, 3 * eff)
        Pd, Pn = yd[:3 * eff, :k] @ yd[:3 * eff, :k].T, yn[:3 * eff, :k] @ yn[:3 * eff, :k].T
        assert np.abs(Pd - Pn).max() < 1e-11                 # same invariant subspace (the gauge fix removes the mixing)
        np.testing.assert_allclose(yd[:3 * eff, :k].T @ yd[:3 * eff, :k], np.eye(k), atol=1e-12)


@pytest.mark.parametrize("kind", ["tight_cluster", "exact_triple", "wide", "negative", "graded"])
@pytest.mark.parametrize("steps", [5, 8, 12, 20, 23, 32])
def test_ritz_fast_path_on_hard_spectra(kind, steps):
    """The tridiagonalisation + bisection + inverse-iteration path of vican_ritz (15 <= n <= 96) on spectra that stress it:
    a cluster of three eigenvalues 1e-13 apart, an exactly triple eigenvalue, a wide range, negative values, a graded
    matrix - eigenvalues and the invariant subspace of the three smallest against LAPACK, orthonormal Ritz vectors.  (Where
    its own validation rejects the result the kernel falls back to the Jacobi iteration: same assertions.)"""
    H, N, g = make_backends(5, 40, 1, 3, 7, np.float64)
    m, ka = 32, 3 * steps
    rng = np.random.default_rng(700 + steps)
    Q = np.linalg.qr(rng.standard_normal((ka, ka)))[0]
    ev = np.sort(rng.uniform(1.0, 50.0, ka))
    if kind == "tight_cluster":
        ev[:3] = 1e-6 * (1.0 + 1e-13 * np.arange(3))
    elif kind == "exact_triple":
        ev[:3] = 0.25
    elif kind == "wide":
        ev[:3] = [1e-9, 3e-7, 2e-3]; ev[-1] = 1e4
    elif kind == "negative":
        ev[:3] = [-3e-8, -1e-9, 4e-7]
    elif kind == "graded":
        ev = np.sort(10.0 ** rng.uniform(-8, 2, ka)); ev[3] = max(ev[3], 50 * ev[2])
    Tm = (Q * ev) @ Q.T
    Tm = 0.5 * (Tm + Tm.T)
    hw = 3 * (m + 1) * 3
    HB = np.zeros((m, hw + 9))
    for j in range(steps):
        kj = 3 * (j + 1)
        HB[j, : kj * 3] = Tm[:kj, 3 * j:3 * j + 3].reshape(-1)
        HB[j, hw:] = np.triu(rng.standard_normal((3, 3)) * 1e-3 + np.eye(3) * 1e-2).reshape(-1)
    Yd, std, gd = H.zeros(3 * (m + 1), 3), H.zeros(16), H.zeros(1, dtype=torch.int32)
    H.ritz(H.from_numpy(HB), hw, steps, 1, 1e-10, 1e-7, -1.0, Yd, std, gd)
    sd, yd = std.cpu().numpy(), Yd.cpu().numpy()[:ka]
    w, U = np.linalg.eigh(Tm)
    nrm = np.linalg.norm(Tm)
    assert np.abs(sd[7:10] - w[:3]).max() <= 4e-15 * nrm                       # three smallest
    assert abs(sd[15] - w[3]) <= 4e-15 * nrm and abs(sd[13] - w[4]) <= 4e-15 * nrm and abs(sd[11] - w[-1]) <= 4e-15 * nrm
    np.testing.assert_allclose(yd.T @ yd, np.eye(3), atol=1e-12)
    assert np.abs(Tm @ yd - yd * sd[7:10]).max() <= 2e-13 * nrm               # Ritz pairs of T itself
    if kind != "graded":
        gap = w[3] - w[2]
        assert np.abs(yd @ yd.T - U[:, :3] @ U[:, :3].T).max() <= 1e-11 * max(1.0, nrm / gap * 1e-3)


def test_gated_launches_are_cancelled_on_the_device():
    """Everything enqueued under a closed gate leaves its outputs untouched; an open gate runs it."""
    H, N, g = make_backends(37, 300, 1, 12, 3, np.float32)
    C, T = 37, 300
    rng = np.random.default_rng(2)
    lamT, cd = H.empty(T, 9), H.empty(C)
    H.init_duals(lamT, cd)
    x = H.from_numpy(np.linalg.qr(rng.standard_normal((3 * C, 3)))[0])
    z_ref = H.zeros(3 * C, 3)
    H.block_op(lamT, x, z_ref)
    gate = H.zeros(1, dtype=torch.int32)
    fx_before = g.fx.clone()
    for val in (0, 2, 1):
        gate.fill_(val)
        z, xp, rc, lamC = H.zeros(3 * C, 3), H.zeros(3 * C, 3), H.zeros(3 * C, 3), H.zeros(C, 9)
        Rt, lamT2 = H.zeros(T, 9), lamT.clone()
        X = H.zeros(3 * C, 3)
        V, Y = H.from_numpy(rng.standard_normal(6 * 3 * C)), H.from_numpy(rng.standard_normal((6, 3)))
        with H.gated(gate):
            H.tall_combine(3 * C, V, 3 * C, 6, Y, X)
            H.gauge_project(x, xp)
            H.block_op(lamT, x, z)
            H.polar_dual(z_ref, rc, lamC, 1)
            H.dual_update(x, Rt, lamT2)
        H.synchronize()
        outs = [X, xp, z, rc, lamC, Rt]
        if val != 1:
            assert all(float(o.abs().max()) == 0.0 for o in outs)
            assert torch.equal(lamT2, lamT) and torch.equal(g.fx, fx_before)
        else:
            assert all(float(o.abs().max()) > 0.0 for o in outs)
            assert torch.equal(z, z_ref) and not torch.equal(lamT2, lamT)
    # outside the context the gate is off again
    gate.fill_(0)
    z = H.zeros(3 * C, 3)
    H.init_duals(lamT, cd)
    H.block_op(lamT, x, z)
    assert torch.equal(z, z_ref)


@pytest.mark.parametrize("cfg", [CONFIGS[1], CONFIGS[4]])
def test_jacobi_scaling_kernels(cfg):
    """vican_jacobi_scale / vican_row_scale / vican_scale_weights: the scaled weights have exactly the row and
    camera sums of w s_c s_t (checked through vican_edge_sums on the chunk layout)."""
    import ctypes as C
    from vican_amd import _lib
    from vican_amd.device import _ptr, _stream
    Cn, T, lo, hi, bt, nwg, er = cfg
    H, N, g = make_backends(Cn, T, lo, hi, 900 + Cn, np.float64, bt, nwg, er)
    deg_t, deg_c = H.empty(T), H.empty(Cn)
    H.trans_degrees(deg_t, deg_c)
    s_t, s_c = H.empty(T), H.empty(Cn)
    H.jacobi_scale(deg_t, s_t); H.jacobi_scale(deg_c, s_c)
    dt, dc = deg_t.cpu().numpy(), deg_c.cpu().numpy()
    np.testing.assert_allclose(s_t.cpu().numpy(), np.where(dt > 0, 1 / np.sqrt(np.where(dt > 0, dt, 1)), 0.0), rtol=1e-15)
    np.testing.assert_allclose(s_c.cpu().numpy(), 1 / np.sqrt(dc), rtol=1e-15)
    x = H.from_numpy(np.random.default_rng(0).standard_normal((T, 3)))
    x0 = x.clone()
    H.row_scale(s_t, x)
    np.testing.assert_allclose(x.cpu().numpy(), x0.cpu().numpy() * s_t.cpu().numpy()[:, None], rtol=1e-15)
    H.set_cg_scaling(s_c, s_t)
    row_sum, cam_sum = H.zeros(T), H.zeros(Cn)
    cam_ws = torch.empty(Cn, dtype=torch.int64, device=H.dev)
    _lib.check(H.lib.vican_edge_sums(C.byref(g.desc), _ptr(H._w_scaled), 1, 1.0, _ptr(row_sum), _ptr(cam_sum), _ptr(cam_ws),
                                     _stream()), "vican_edge_sums")
    wt = N.w * s_c.cpu().numpy()[N.col] * s_t.cpu().numpy()[N.row]
    assert wt.max() <= 1.0 + 1e-12
    er_, ec_ = np.zeros(T), np.zeros(Cn)
    np.add.at(er_, N.row, wt); np.add.at(ec_, N.col, wt)
    np.testing.assert_allclose(row_sum.cpu().numpy(), er_, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(cam_sum.cpu().numpy(), ec_, rtol=1e-12, atol=1e-13)
    assert H._cg_w is H._w_scaled and H._cg_wmax == 1.0
    H.clear_cg_scaling()
    assert H._cg_w is g.w


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_fused_dual_update_redo_path(dt):
    """The 12-wavefront fused dual-update sweep carries no SVD: a row whose Newton polar iteration does not apply raises
    the redo word and the gated 8-wavefront kernel (with the SVD path) reruns the sweep (vican_wsweep.hip, FB).  Rows of
    one edge with an ill-conditioned camera operand (cond 5e3) force that path: the result is bit-identical to the same
    graph planned for 8 wavefronts (single launch, SVD in the kernel), the redo word is cleared afterwards, and a second
    call with well-conditioned operands (no redo) is again identical between the two plans."""
    C, T = 100, 3000
    H12, N, g12 = make_backends(C, T, 1, 24, 77, dt, "wave12", 6)
    H8, _, g8 = make_backends(C, T, 1, 24, 77, dt, "wave8", 6)
    assert g12.wg_waves == 12 and g8.wg_waves == 8
    rng = np.random.default_rng(8)
    u, v = synth.random_rotations(rng, C), synth.random_rotations(rng, C)
    # per camera U_c diag(1, 1, 2e-4) V_c^T: |.|_F < sqrt 3, normalised det 3.7e-4 < 1e-3 - rows of one edge are
    # ill-conditioned (but 1e3 above the f32 products' rounding), sums over several cameras are not
    bad = ((u * np.array([1.0, 1.0, 2e-4])) @ np.swapaxes(v, 1, 2)).reshape(3 * C, 3)
    good = synth.random_rotations(rng, C).reshape(3 * C, 3)
    for rc, redo in ((bad, True), (good, False)):
        outs = []
        for K, g in ((H12, g12), (H8, g8)):
            lam0, cd = K.empty(T, 9), K.empty(C)
            K.init_duals(lam0, cd)
            Rt, L, z = K.zeros(T, 9), K.zeros(T, 9), K.empty(3 * C, 3)
            K.dual_update_op(K.from_numpy(rc), Rt, L, z)
            torch.cuda.synchronize()
            assert int(g.fx.view(torch.int32)[2 * 12 + 4]) == 0         # redo word re-armed by dual_svd_kernel
            outs.append((Rt, L, z))
        for a, b in zip(outs[0], outs[1]):
            assert torch.equal(a, b)
        Rn, Ln, zn = N.zeros(T, 9), N.zeros(T, 9), N.empty(3 * C, 3)
        N.dual_update_op(N.from_numpy(rc), Rn, Ln, zn)
        scale = np.abs(zn.numpy()).max()
        tol = (1e-7 if dt == np.float64 else 1e-4) if redo else (1e-9 if dt == np.float64 else 3e-6)
        assert np.abs(outs[0][2].cpu().numpy() - zn.numpy()).max() <= tol * scale
    # the rows that forced the redo exist: rows of one edge
    assert (np.diff(N.row_ptr) == 1).sum() > 50


@pytest.mark.parametrize("cfg", CONFIGS)
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_fused_dual_update_op(cfg, dt):
    """vican_dual_update_op: the outputs of vican_dual_update, and z_raw = P_new rc from the same pass (Newton polar
    factors inside the sweep) - against the unfused path on the same device and the NumPy mirror."""
    C, T, lo, hi, bt, nwg, er = cfg
    H, N, g = make_backends(C, T, lo, hi, 500 + C, dt, bt, nwg, er)
    rng = np.random.default_rng(6)
    lam0, cd = H.empty(T, 9), H.empty(C)
    H.init_duals(lam0, cd)
    rc = synth.random_rotations(rng, C).reshape(3 * C, 3)
    Rt1, L1, Rt2, L2 = H.zeros(T, 9), H.zeros(T, 9), H.zeros(T, 9), H.zeros(T, 9)
    zraw, zref = H.empty(3 * C, 3), H.empty(3 * C, 3)
    H.dual_update_op(H.from_numpy(rc), Rt1, L1, zraw)
    H.dual_update(H.from_numpy(rc), Rt2, L2)
    nn = lambda t: torch.nan_to_num(t, nan=-7.0, posinf=-8.0, neginf=-9.0)      # rows without edges carry inf/nan duals
    assert torch.equal(nn(Rt1), nn(Rt2)) and torch.equal(nn(L1), nn(L2))      # same exact row sums, same SVD kernel
    nz = np.diff(N.row_ptr) > 0
    well = nz & (np.diff(N.row_ptr) >= 2)
    # unfused reference on the device: block_op with the new duals (rows without edges carry inf/nan duals: mask them)
    Lc = L2.clone(); Lc[torch.from_numpy(~nz).to(Lc.device)] = 0.0
    H.set_duals(Lc); H.block_op(Lc, H.from_numpy(rc), zref)
    Rn, Ln, zn = N.zeros(T, 9), N.zeros(T, 9), N.empty(3 * C, 3)
    N.dual_update_op(N.from_numpy(rc), Rn, Ln, zn)
    scale = np.abs(zn.numpy()).max()
    tol = 1e-9 if dt == np.float64 else 3e-6
    assert np.abs(zraw.cpu().numpy() - zn.numpy()).max() <= tol * scale
    # (the unfused product lamT_new (Z_t beta) loses cond(Z_t) digits on rows with one or two noisy blocks: sanity level only)
    assert np.abs(zraw.cpu().numpy() - zref.cpu().numpy()).max() <= (1e-6 if dt == np.float64 else 1e-4) * scale
    assert np.abs(Rt1.cpu().numpy()[well] - Rn.numpy()[well]).max() < (1e-9 if dt == np.float64 else 2e-3)
    # bit-identical on repeats (exact integer sums, deterministic Newton iteration)
    z2 = H.empty(3 * C, 3)
    H.dual_update_op(H.from_numpy(rc), Rt2, L2, z2)
    assert torch.equal(zraw, z2)
    # start-block normalisation: z_raw beta^-1
    beta = np.triu(rng.standard_normal((3, 3))) + 3 * np.eye(3)
    zs = H.empty(3 * C, 3)
    H.right_solve3(zraw, H.from_numpy(beta.reshape(-1)), zs)
    assert np.abs(zs.cpu().numpy() - zraw.cpu().numpy() @ np.linalg.inv(beta)).max() <= 1e-12 * scale
    beta[1, 1] = 0.0                                                  # dropped column
    H.right_solve3(zraw, H.from_numpy(beta.reshape(-1)), zs)
    assert not zs.cpu().numpy()[:, 1].any()


def test_newton_polar_inside_the_sweep_handles_reflections_and_singular_rows():
    """Rows whose block sum Z_t has a negative determinant (polar factor = reflection, as U V^T of the reference's
    SVD) or is singular (SVD fallback) - through a graph with one edge per row and hand-made blocks."""
    from vican_amd.device import HipBackend, LocalGraph
    rng = np.random.default_rng(8)
    C, T = 4, 64
    rp = np.arange(T + 1, dtype=np.int32)
    col = (np.arange(T) % C).astype(np.int32)
    blk = rng.standard_normal((T, 3, 3))
    blk[::3] *= -np.sign(np.linalg.det(blk[::3]))[:, None, None]     # det < 0
    blk[5] = np.outer([1.0, 2.0, 3.0], [0.5, -1.0, 2.0])             # rank 1
    blk[7] = 0.0                                                     # zero block
    blk[9] = np.diag([1.0, 1e-11, 1.0])                              # cond 1e11
    a = np.ones(T)
    dev = torch.device("cuda:0")
    g = LocalGraph(C, torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev), torch.from_numpy(blk.reshape(T, 9)).to(dev),
                   torch.from_numpy(a).to(dev))
    H = HipBackend(g)
    lam0, cd = H.empty(T, 9), H.empty(C)
    H.init_duals(lam0, cd)
    rc = np.tile(np.eye(3), (C, 1))                                   # Z_t = M_t^T
    Rt, L, zraw = H.zeros(T, 9), H.zeros(T, 9), H.empty(3 * C, 3)
    H.dual_update_op(H.from_numpy(rc), Rt, L, zraw)
    u, s, vt = np.linalg.svd(np.swapaxes(blk, 1, 2))
    w = u @ vt
    good = np.ones(T, bool); good[[5, 7]] = False                     # polar factor not unique there: only finiteness
    z = np.zeros((C, 3, 3))
    np.add.at(z, col[good], blk[good] @ w[good])
    zb = np.zeros((C, 3, 3))
    np.add.at(zb, col[~good], np.abs(blk[~good]).sum(axis=(1, 2))[:, None, None] * np.ones((3, 3)))
    got = zraw.cpu().numpy().reshape(C, 3, 3)
    assert np.isfinite(got).all()
    assert (np.abs(got - z) <= 1e-9 * np.abs(z).max() + 2.0 * zb).all()


@pytest.mark.parametrize("n", [3, 15, 1020, 3000, 16384])
def test_lanczos_seed_matches_the_launch_sequence(n):
    """vican_lanczos_seed (one launch) against rows_to_cols + tall_gram + chol_qr3 + right_solve3 and NumPy."""
    H, N, g = make_backends(5, 40, 1, 3, 7, np.float64)
    rng = np.random.default_rng(n)
    x0 = rng.standard_normal((n, 3)) @ np.array([[2.0, 0.3, -0.1], [0.0, 1.5, 0.4], [0.0, 0.0, 0.7]])
    zraw = rng.standard_normal((n, 3))
    m = 2
    outs = []
    for K, fused in ((H, True), (H, False), (N, True)):
        V, beta, xrow, z = K.zeros(3 * (m + 1) * n), K.zeros(9), K.zeros(n, 3), K.zeros(n, 3)
        if fused:
            assert K.lanczos_seed(K.from_numpy(x0), V, n, beta, xrow, K.from_numpy(zraw), z)
        else:
            R, G = K.zeros(3 * n), K.zeros(9)
            K.rows_to_cols(n, K.from_numpy(x0), R, n, 0)
            K.tall_gram(n, R, n, 3, R, G)
            K.chol_qr3(n, R, G, V, n, 0, beta, xrow, 0.0)
            K.right_solve3(K.from_numpy(zraw), beta, z)
        outs.append([t.cpu().numpy().copy() for t in (V, beta, xrow, z)])
    q, r = np.linalg.qr(x0)
    sgn = np.sign(np.diag(r))
    for o in outs:
        assert np.abs(o[1].reshape(3, 3) - r * sgn[:, None]).max() < 1e-10 * np.abs(r).max()       # beta = R factor with positive diagonal
        assert np.abs(o[2] - q * sgn[None, :]).max() < 1e-10
        assert np.abs(o[0][: 3 * n].reshape(3, n).T - o[2]).max() == 0.0                           # basis columns = the same block
        assert np.abs(o[3] - zraw @ np.linalg.inv(r * sgn[:, None])).max() < 1e-9 * np.abs(o[3]).max()
    for a, b in zip(outs[0], outs[1]):
        assert np.abs(a - b).max() < 1e-12 * max(1.0, np.abs(b).max())
    # a zero column of the start block: zero pivot -> zero column (pivot rule of vican_chol_qr3 with pivot_floor 0)
    x1 = x0.copy(); x1[:, 2] = 0.0
    V, beta, xrow = H.zeros(3 * (m + 1) * n), H.zeros(9), H.zeros(n, 3)
    H.lanczos_seed(H.from_numpy(x1), V, n, beta, xrow)
    assert beta.cpu().numpy()[8] == 0.0 and not xrow.cpu().numpy()[:, 2].any() and np.isfinite(xrow.cpu().numpy()).all()


@pytest.mark.parametrize("cfg", [CONFIGS[1], CONFIGS[2], CONFIGS[4], CONFIGS[6], CONFIGS[8], CONFIGS[10], CONFIGS[12]])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_lanczos_step_folding_the_slabs_itself(cfg, dt):
    """lanczos_cam_step(from_slabs=True) - the cooperative kernel reads z from the sweep's fixed-point slabs - gives
    bit-identical results to slab fold + step (same exact integer sums, same conversion)."""
    C, T, lo, hi, bt, nwg, er = cfg
    H, N, g = make_backends(C, T, lo, hi, 700 + C, dt, bt, nwg, er)
    rng = np.random.default_rng(9)
    n, j, m = 3 * C, 1, 4
    Q = np.linalg.qr(rng.standard_normal((n, 3 * (j + 1))))[0]
    V0 = np.zeros((3 * (m + 1), n)); V0[: 3 * (j + 1)] = Q.T
    lamT, cd = H.empty(T, 9), H.empty(C)
    H.init_duals(lamT, cd)
    lamC = H.empty(C, 9); H.scaled_identity(cd, lamC)
    x = H.from_numpy(np.ascontiguousarray(Q[:, 3 * j: 3 * j + 3]))
    outs = []
    for from_slabs in (False, True):
        V = H.from_numpy(V0.reshape(-1).copy())
        R, Hs, G = H.empty(3 * n), H.empty(3 * (m + 1) * 3), H.empty(9)
        Hcol, beta, xo, z = H.zeros(3 * (m + 1) * 3), H.zeros(9), H.empty(n, 3), H.zeros(n, 3)
        if from_slabs:
            H.block_op_slabs(lamT, x)
            H.lanczos_cam_step(lamC, V, n, j, z, R, Hs, G, Hcol, beta, xo, 0.0, from_slabs=True)
        else:
            H.block_op(lamT, x, z)
            H.lanczos_cam_step(lamC, V, n, j, z, R, Hs, G, Hcol, beta, xo, 0.0)
        outs.append([t.clone() for t in (Hcol, beta, xo, V)])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("cfg", [c for c in CONFIGS if isinstance(c[4], str)])
def test_wave_layout_carries_its_index_in_two_bytes(cfg):
    """vican_graph_t.idx16 (vican_pack_idx16): camera | row << 10 of every slot of a wave-layout graph, 0xFFFF for padding -
    the word the rotation sweeps stream instead of the 4-byte idx."""
    C_, T, lo, hi, bt, n_wg, empty = cfg
    H, N, g = make_backends(C_, T, lo, hi, 13, np.float32, bt, n_wg, empty)
    assert g.layout == "wave" and g.idx16 is not None and g.desc.idx16 == g.idx16.data_ptr()
    idx = g.idx.cpu().numpy().view(np.uint32)
    i16 = g.idx16.cpu().numpy().view(np.uint16)
    pad = idx == 0xFFFFFFFF
    assert np.array_equal(i16[pad], np.full(int(pad.sum()), 0xFFFF, dtype=np.uint16))
    cam, row = idx[~pad] & 0xFFFF, idx[~pad] >> 16
    assert cam.max() < 1024 and row.max() < 64 and not np.any((cam == 1023) & (row == 63))
    assert np.array_equal(i16[~pad], (cam | (row << 10)).astype(np.uint16))


def test_float32_weight_stream_of_the_one_row_cg_product_is_exact(monkeypatch):
    """vican_graph_t.w32: one-row wave graphs whose translation weights are all float32 values (every dtype=float32 problem)
    stream them as 4 bytes in plain slot order; the CG solve is bit-identical to the float64 stream (VICAN_CG_W32=0), scaled
    weights (another array than the one w32 mirrors) take the float64 stream, and weights that are not float32 values get no
    copy at all."""
    from vican_amd.device import HipBackend, LocalGraph
    from vican_amd.solver import Comm, TranslationSolver
    C_, T_ = 300, 900
    rp, col, blk, a, w, u, v = random_graph(C_, T_, 200, 256, 31, False)                 # one row per 256-slot chunk
    w32ok = w.astype(np.float32).astype(np.float64)
    dev = torch.device("cuda:0")
    to = lambda x, d=None: torch.from_numpy(x).to(dev) if d is None else torch.from_numpy(x).to(dev, d)
    rng = np.random.default_rng(5)
    rc = np.linalg.qr(rng.standard_normal((C_, 3, 3)))[0].reshape(3 * C_, 3)
    rt = np.linalg.qr(rng.standard_normal((T_, 3, 3)))[0].reshape(T_, 9)
    outs = {}
    for tag, weights, env in (("w32", w32ok, "1"), ("f64", w32ok, "0"), ("inexact", w, "1")):
        monkeypatch.setenv("VICAN_CG_W32", env)
        g = LocalGraph(C_, to(rp), to(col), to(blk, torch.float32), to(a, torch.float32), to(weights), to(u), to(v), layout="wave")
        assert g.n_chunk == g.n_time and (g.w32 is not None) == (tag == "w32")
        if g.w32 is not None:
            assert g.desc.w32 == g.w32.data_ptr() and g.desc.w32_src == g.w.data_ptr()
        K = HipBackend(g)
        ts = TranslationSolver(K, Comm.single(), rtol=1e-9)
        ts.setup(K.from_numpy(rc), K.from_numpy(rt))
        xc, xt = ts.solve(3 * (C_ + T_))
        outs[tag] = (xc.clone(), xt[:T_].clone(), ts.info["cg_iters"])
        if tag == "w32":                                            # tight mode scales the weights: float64 stream, still a solve
            from vican_amd.solver import TightTranslationSolver
            tt = TightTranslationSolver(K, Comm.single(), rtol=1e-10)
            tt.setup(K.from_numpy(rc), K.from_numpy(rt))
            xc2, _ = tt.solve(3 * (C_ + T_))
            assert tt.info["converged"] and float((xc2 - xc).abs().max()) < 1e-5 * max(float(xc.abs().max()), 1.0)
    assert outs["w32"][2] == outs["f64"][2] and torch.equal(outs["w32"][0], outs["f64"][0]) and torch.equal(outs["w32"][1], outs["f64"][1])
