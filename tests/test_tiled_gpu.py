"""More cameras than the LDS-resident sweeps hold (C > 1024; the reference has no limit, bipgo.py:225-232): the
camera-tiled path (device.TiledGraph / TiledBackend) against the NumPy restatement at C = 1500, and - with the tile
size forced down - against the goldens of the real reference."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import golden_cases as gc                                                   # noqa: E402
from vican_amd.backend_cpu import NumpyBackend                                      # noqa: E402
from test_kernels_gpu import random_graph                                   # noqa: E402
from util import expected, load_golden, pose_errors, rebuild_inputs, translation_tol   # noqa: E402
from vican_amd.solver import Comm, solve_on_backend                         # noqa: E402


@pytest.fixture(params=["fused", "wave", "block"], autouse=True)
def tile_layout(request, monkeypatch):
    """Every test of this file runs three ways: "fused" = the default (wave-layout tiles with a SHARED chunking, the operator as
    one launch that reads every block once: vican_tiled_op), "wave" = wave-layout tiles with their own chunkings (a rows pass and
    a camera pass per tile: vican_tile_rows / vican_tile_cams, the wave CG product in partial mode), "block" = block-layout tiles
    (vican_bip_apply twice, the block CG product)."""
    if request.param == "fused":
        monkeypatch.setenv("VICAN_TILE_LAYOUT", "wave")
    else:
        monkeypatch.setenv("VICAN_TILE_LAYOUT", request.param)
        monkeypatch.setenv("VICAN_TILE_SHARED", "0")
    return request.param


def _tiled(C, rp, col, blk, a, w, u, v, dt, tile):
    from vican_amd.device import TiledBackend, TiledGraph
    dev = torch.device("cuda:0")
    tdt = torch.float32 if dt == np.float32 else torch.float64
    g = TiledGraph(C, torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev), torch.from_numpy(blk).to(dev, tdt),
                   torch.from_numpy(a).to(dev, tdt), torch.from_numpy(w).to(dev), torch.from_numpy(u).to(dev), torch.from_numpy(v).to(dev),
                   tile=tile)
    import os
    assert all(t.layout == os.environ["VICAN_TILE_LAYOUT"] for t in g.tiles)
    K = TiledBackend(g)
    shared = os.environ.get("VICAN_TILE_SHARED", "1") != "0"
    assert (g.shared_chunks is not None) == shared and (K._fused is not None) == shared, K.coop_failures
    K._expect_fused = shared
    return g, K


@pytest.mark.parametrize("C,T,tile", [(1500, 400, 1024), (90, 300, 32)])
def test_tiled_operator_and_dual_update_match_numpy(C, T, tile):
    rp, col, blk, a, w, u, v = random_graph(C, T, 3, 9, 5, False)
    g, K = _tiled(C, rp, col, blk, a, w, u, v, np.float64, tile)
    N = NumpyBackend(C, rp, col, blk, a, w, u, v, storage=np.float64)
    assert len(g.tiles) == -(-C // tile)
    rng = np.random.default_rng(1)
    x = np.linalg.qr(rng.standard_normal((3 * C, 3)))[0]
    lh, ch, ln, cn = K.empty(T, 9), K.empty(C), N.empty(T, 9), N.empty(C)
    K.init_duals(lh, ch); N.init_duals(ln, cn)
    assert np.abs(lh.cpu().numpy() - ln.numpy()).max() < 1e-12 and np.abs(ch.cpu().numpy() - cn.numpy()).max() < 1e-10
    zh, zn = K.empty(3 * C, 3), N.empty(3 * C, 3)
    K.block_op(lh, K.from_numpy(x), zh); N.block_op(ln, N.from_numpy(x), zn)
    assert np.abs(zh.cpu().numpy() - zn.numpy()).max() <= 1e-10 * np.abs(zn.numpy()).max()
    rc = np.linalg.qr(rng.standard_normal((C, 3, 3)))[0].reshape(3 * C, 3)
    Rh, Rn = K.empty(T, 9), N.empty(T, 9)
    K.dual_update(K.from_numpy(rc), Rh, lh); N.dual_update(N.from_numpy(rc), Rn, ln)
    assert np.abs(Rh.cpu().numpy() - Rn.numpy()).max() < 1e-9
    assert np.abs(lh.cpu().numpy() - ln.numpy()).max() <= 1e-9 * np.abs(ln.numpy()).max()
    for _ in range(3):                                                               # with the new (full 3x3) duals; alternating share buffers
        K.block_op(lh, K.from_numpy(x), zh); N.block_op(ln, N.from_numpy(x), zn)
        assert np.abs(zh.cpu().numpy() - zn.numpy()).max() <= 1e-9 * np.abs(zn.numpy()).max()
    assert (K._fused is not None) == K._expect_fused, K.coop_failures             # the fused launch was not refused


def test_full_solve_with_1500_cameras_matches_numpy():
    """A consistent synthetic scene (1500 cameras, 2500 timesteps, 4 cameras per timestep, 1e-3 noise) through the whole
    solve - block Lanczos on 4500-row vectors, tiled operator and dual updates, tiled right-hand side, tiled CG product
    - against the NumPy restatement of the same solver."""
    from vican_amd import synth
    from vican_amd.device import TiledBackend, TiledGraph
    C, T = 1500, 2500
    dev = torch.device("cuda:0")
    gr = synth.make_merged_graph_torch(C, T, 4, dev, torch.float64, seed=3, sigma_r=1e-3, sigma_t=1e-3)
    g = TiledGraph(C, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"], tile=1024)
    K = TiledBackend(g)
    assert len(g.tiles) == 2
    h = {k: gr[k].cpu().numpy() for k in ("row_ptr", "col", "blk", "a", "w", "u", "v")}
    N = NumpyBackend(C, h["row_ptr"], h["col"], h["blk"], h["a"], h["w"], h["u"], h["v"], storage=np.float64)
    out_h = solve_on_backend(K, Comm(), 4, 3 * (C + T))
    out_n = solve_on_backend(N, Comm(), 4, 3 * (C + T))
    assert np.abs(out_h[0].cpu().numpy() - out_n[0].numpy()).max() < 1e-7          # camera rotations
    assert np.abs(out_h[1].cpu().numpy()[:T] - out_n[1].numpy()[:T]).max() < 1e-7  # timestep rotations
    assert abs(out_h[4]["cg_iters"] - out_n[4]["cg_iters"]) <= 3
    scale = np.abs(out_n[2].numpy()).max()
    assert np.abs(out_h[2].cpu().numpy() - out_n[2].numpy()).max() < 1e-3 * scale     # loosely converged CG (rtol 1e-5)
    # ... and close to the ground truth of the scene (gauge: camera 0 at the identity)
    Rc = out_h[0].cpu().numpy().reshape(C, 3, 3).transpose(0, 2, 1)
    gt = gr["R_cam"].cpu().numpy()
    rel = np.einsum("ij,cjk->cik", gt[0].T, gt)
    assert np.abs(Rc - rel).max() < 5e-2


@pytest.mark.parametrize("name,dt", [("g3_medium", "float64"), ("g2_small", "float32")])
def test_dropin_on_forced_tiles_matches_reference(name, dt, monkeypatch):
    """VICAN_TILE_CAMS forces the tiled path on a golden case (40 cameras in tiles of 16): same answer as the reference."""
    from vican.bipgo import bipartite_se3sync
    monkeypatch.setenv("VICAN_TILE_CAMS", "16" if name == "g3_medium" else "3")
    g = load_golden(name)
    case, src, cons, (nr, nt, ff) = rebuild_inputs(name, g)
    exp = expected(g, "conjugate_gradient", dt)
    info = {}
    res = bipartite_se3sync(src, constraints=cons, noise_model_r=nr, noise_model_t=nt, edge_filter=ff, maxiter=gc.MAXITER,
                            lsqr_solver="conjugate_gradient", dtype=np.dtype(dt).type, info=info)
    assert info["layout"] == "tiled"
    rot, tr = pose_errors(res, exp)
    assert rot < (1e-7 if dt == "float64" else 5e-6), rot
    # (f32 blocks: the tiles' partial sums round differently from the fused sweep: 3e-8 rad in the rotations, 1e-6 m here)
    assert tr < max(translation_tol(name, dt), 5e-6 if dt == "float32" else 0.0), tr
    # lsqr_solver="direct" on the tiles (round 2 refused it): the reference's LSQR golden
    expd = expected(g, "direct", dt)
    if expd:
        info2 = {}
        res2 = bipartite_se3sync(src, cons, nr, nt, ff, gc.MAXITER, "direct", np.dtype(dt).type, info=info2)
        rot2, tr2 = pose_errors(res2, expd)
        assert info2["layout"] == "tiled" and info2["lsqr_istop"] in (1, 2)
        assert rot2 < (1e-7 if dt == "float64" else 5e-6) and tr2 < (2e-6 if dt == "float64" else 5e-4), (rot2, tr2)


@pytest.mark.parametrize("C,T,tile", [(4000, 3000, 1024), (70, 500, 16)])
def test_tiled_translation_stage_has_no_camera_limit(C, T, tile):
    """The CG product tile by tile (vican_cg_sweep_partial + vican_cg_combine_rows): right-hand side and CG on 4000
    cameras - beyond what any single LDS table holds (round 2 stopped at ~3300) - against the NumPy restatement, plain
    and in tight (Jacobi-scaled) mode."""
    from vican_amd.solver import TightTranslationSolver, TranslationSolver
    rp, col, blk, a, w, u, v = random_graph(C, T, 3, 9, 11, False)
    g, K = _tiled(C, rp, col, blk, a, w, u, v, np.float64, tile)
    N = NumpyBackend(C, rp, col, blk, a, w, u, v, storage=np.float64)
    rng = np.random.default_rng(2)
    rc = np.linalg.qr(rng.standard_normal((C, 3, 3)))[0].reshape(3 * C, 3)
    rt = np.linalg.qr(rng.standard_normal((T, 3, 3)))[0].reshape(T, 9)
    for cls, rtol, tol in ((TranslationSolver, 1e-9, 1e-6), (TightTranslationSolver, 1e-10, 1e-6)):
        th, tn = cls(K, Comm.single(), rtol=rtol), cls(N, Comm.single(), rtol=rtol)
        th.setup(K.from_numpy(rc), K.from_numpy(rt)); tn.setup(N.from_numpy(rc), N.from_numpy(rt))
        assert np.abs(th.b_c.cpu().numpy() - tn.b_c.numpy()).max() <= 1e-11 * np.abs(tn.b_c.numpy()).max()
        xh = [x.cpu().numpy().copy() for x in th.solve(3 * (C + T))]
        xn = [x.numpy().copy() for x in tn.solve(3 * (C + T))]
        assert th.info["converged"] and abs(th.info["cg_iters"] - tn.info["cg_iters"]) <= 2
        scale = max(np.abs(xn[0]).max(), 1.0)
        assert np.abs(xh[0] - xn[0]).max() < tol * scale and np.abs(xh[1][:T] - xn[1][:T]).max() < tol * scale


def test_tile_layout_default_is_wave_with_block_fallback(monkeypatch):
    """Without VICAN_TILE_LAYOUT a tile takes the wave layout if its rows fit a 64-lane chunk (128 f64 edges) and the block
    layout otherwise - here 400 cameras in two tiles of 200, 300-380 cameras per timestep with 60 % of the second tile's edges
    dropped: the first tile's rows have ~150-190 edges (block), the second tile's ~50-90 (wave); the mixed operator against the
    NumPy restatement."""
    from vican_amd.device import TiledBackend, TiledGraph
    monkeypatch.delenv("VICAN_TILE_LAYOUT")
    C, T, tile = 400, 60, 256
    rp, col, blk, a, w, u, v = random_graph(C, T, 300, 380, 4, False)
    keep = (col < 200) | (np.random.default_rng(9).random(len(col)) < 0.4)
    rows = np.repeat(np.arange(T), np.diff(rp))[keep]
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=T))]).astype(np.int32)
    col, blk, a, w, u, v = col[keep], blk[keep], a[keep], w[keep], u[keep], v[keep]
    dev = torch.device("cuda:0")
    g = TiledGraph(C, torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev), torch.from_numpy(blk).to(dev), torch.from_numpy(a).to(dev),
                   torch.from_numpy(w).to(dev), torch.from_numpy(u).to(dev), torch.from_numpy(v).to(dev), tile=tile)
    assert [t.layout for t in g.tiles] == ["block", "wave"]
    K = TiledBackend(g)
    N = NumpyBackend(C, rp, col, blk, a, w, u, v, storage=np.float64)
    rng = np.random.default_rng(1)
    x = np.linalg.qr(rng.standard_normal((3 * C, 3)))[0]
    lh, ch, ln, cn = K.empty(T, 9), K.empty(C), N.empty(T, 9), N.empty(C)
    K.init_duals(lh, ch); N.init_duals(ln, cn)
    rc = np.linalg.qr(rng.standard_normal((C, 3, 3)))[0].reshape(3 * C, 3)
    Rh, Rn = K.empty(T, 9), N.empty(T, 9)
    K.dual_update(K.from_numpy(rc), Rh, lh); N.dual_update(N.from_numpy(rc), Rn, ln)
    assert np.abs(Rh.cpu().numpy() - Rn.numpy()).max() < 1e-9
    zh, zn = K.empty(3 * C, 3), N.empty(3 * C, 3)
    K.block_op(lh, K.from_numpy(x), zh); N.block_op(ln, N.from_numpy(x), zn)
    assert np.abs(zh.cpu().numpy() - zn.numpy()).max() <= 1e-9 * np.abs(zn.numpy()).max()


def test_tiled_dense_rows_take_the_one_row_kernels(tile_layout):
    """Tiles whose rows fill a chunk on their own (600 cameras in tiles of 300, 360-440 cameras per timestep, f32 blocks: ~200
    edges per tile row, one row per 256-slot chunk, 12 wavefronts): the accumulator-free one-row instantiations of the rows
    pass (MODE 1) and the camera pass (MODE 4), and the general CG product in partial mode (the one-row CG kernel has none)."""
    from vican_amd.solver import TranslationSolver
    C, T, tile = 600, 4000, 300
    rp, col, blk, a, w, u, v = random_graph(C, T, 360, 440, 6, False)
    g, K = _tiled(C, rp, col, blk, a, w, u, v, np.float32, tile)
    if tile_layout in ("wave", "fused"):
        assert all(t.n_chunk == t.n_time and t.wg_waves == 12 for t in g.tiles)
    N = NumpyBackend(C, rp, col, blk, a, w, u, v, storage=np.float32)
    rng = np.random.default_rng(1)
    x = np.linalg.qr(rng.standard_normal((3 * C, 3)))[0]
    lh, ch, ln, cn = K.empty(T, 9), K.empty(C), N.empty(T, 9), N.empty(C)
    K.init_duals(lh, ch); N.init_duals(ln, cn)
    zh, zn = K.empty(3 * C, 3), N.empty(3 * C, 3)
    K.block_op(lh, K.from_numpy(x), zh); N.block_op(ln, N.from_numpy(x), zn)                # scaled-identity duals
    assert np.abs(zh.cpu().numpy() - zn.numpy()).max() <= 2e-5 * np.abs(zn.numpy()).max()
    rc = np.linalg.qr(rng.standard_normal((C, 3, 3)))[0].reshape(3 * C, 3)
    Rh, Rn = K.empty(T, 9), N.empty(T, 9)
    K.dual_update(K.from_numpy(rc), Rh, lh); N.dual_update(N.from_numpy(rc), Rn, ln)
    assert np.abs(Rh.cpu().numpy() - Rn.numpy()).max() < 2e-5
    # (sums of ~400 random rotations: some Z_t are nearly singular and U S^-1 U^T amplifies the float32 rounding of the blocks -
    #  the operator is compared on the SAME full 3x3 duals, the restatement's)
    lh = K.from_numpy(ln.numpy())
    K.set_duals(lh)
    K.block_op(lh, K.from_numpy(x), zh); N.block_op(ln, N.from_numpy(x), zn)
    assert np.abs(zh.cpu().numpy() - zn.numpy()).max() <= 2e-5 * np.abs(zn.numpy()).max()
    rt = np.linalg.qr(rng.standard_normal((T, 3, 3)))[0].reshape(T, 9)
    th, tn = TranslationSolver(K, Comm.single(), rtol=1e-9), TranslationSolver(N, Comm.single(), rtol=1e-9)
    th.setup(K.from_numpy(rc), K.from_numpy(rt)); tn.setup(N.from_numpy(rc), N.from_numpy(rt))
    xh = [x_.cpu().numpy().copy() for x_ in th.solve(3 * (C + T))]
    xn = [x_.numpy().copy() for x_ in tn.solve(3 * (C + T))]
    assert th.info["converged"] and abs(th.info["cg_iters"] - tn.info["cg_iters"]) <= 2
    scale = max(np.abs(xn[0]).max(), 1.0)
    assert np.abs(xh[0] - xn[0]).max() < 1e-6 * scale and np.abs(xh[1][:T] - xn[1][:T]).max() < 1e-6 * scale


def test_rows_packed_for_the_shared_chunking_give_the_same_answers(tile_layout):
    """TiledGraph(permute_rows=True) - the rows in the order vican_plan_rows_multi packs them for the shared chunking, what
    device.make_backend builds - against the rows in their own order: fewer slots; the operator (a sum over rows) and the per-row
    results after `unpermute_rows` the same up to the f32 rounding of the lanes' partial sums (which edges of a row share a lane
    depends on the chunking), the translation stage the same iterate."""
    if tile_layout != "fused":
        pytest.skip("the packing serves the shared chunking")
    from vican_amd.device import TiledBackend, TiledGraph
    from vican_amd.solver import TranslationSolver
    C, T, tile = 800, 1200, 200                                     # 4 tiles, rows of 62.5 +- 6 edges per tile: 4 rows of 256 slots at best, 3 consecutive ones most of the time
    rp, col, blk, a, w, u, v = random_graph(C, T, 246, 254, 21, False)
    dev = torch.device("cuda:0")
    to = lambda x, d=None: torch.from_numpy(x).to(dev) if d is None else torch.from_numpy(x).to(dev, d)
    args = (C, to(rp), to(col), to(blk, torch.float32), to(a, torch.float32), to(w), to(u), to(v))
    g0, g1 = TiledGraph(*args, tile=tile), TiledGraph(*args, tile=tile, permute_rows=True)
    assert g1.row_perm is not None and g0.row_perm is None
    assert sorted(g1.row_perm.cpu().tolist()) == list(range(T))
    assert g1.padded_slots() < 0.9 * g0.padded_slots(), (g1.padded_slots(), g0.padded_slots())
    K0, K1 = TiledBackend(g0), TiledBackend(g1)
    assert K0._fused is not None and K1._fused is not None
    rng = np.random.default_rng(4)
    x = K0.from_numpy(np.linalg.qr(rng.standard_normal((3 * C, 3)))[0])
    l0, c0, l1, c1 = K0.empty(T, 9), K0.empty(C), K1.empty(T, 9), K1.empty(C)
    K0.init_duals(l0, c0); K1.init_duals(l1, c1)
    close = lambda p, q, tol=2e-6: float((p - q).abs().max()) <= tol * max(float(q.abs().max()), 1e-30)
    assert close(g1.unpermute_rows(l1), l0, 1e-12) and close(c0, c1, 1e-12)
    z0, z1 = K0.empty(3 * C, 3), K1.empty(3 * C, 3)
    K0.block_op(l0, x, z0); K1.block_op(l1, x, z1)
    assert close(z1, z0)
    rc = K0.from_numpy(np.linalg.qr(rng.standard_normal((C, 3, 3)))[0].reshape(3 * C, 3))
    R0, R1 = K0.empty(T, 9), K1.empty(T, 9)
    K0.dual_update(rc, R0, l0); K1.dual_update(rc, R1, l1)
    assert close(g1.unpermute_rows(R1), R0) and close(g1.unpermute_rows(l1), l0, 1e-4)
    K0.block_op(l0, x, z0); K1.block_op(l1, x, z1)
    assert close(z1, z0, 1e-4)
    outs = []
    for K, g in ((K0, g0), (K1, g1)):
        ts = TranslationSolver(K, Comm.single(), rtol=1e-8)
        ts.setup(rc, R0 if g is g0 else g1.permute_rows(R0))
        xc, xt = ts.solve(3 * (C + T))
        outs.append((xc.clone(), g.unpermute_rows(xt[:T]).clone() if g is g1 else xt[:T].clone(), ts.info["cg_iters"]))
    assert abs(outs[0][2] - outs[1][2]) <= 1
    scale = max(float(outs[0][1].abs().max()), 1.0)
    assert float((outs[0][0] - outs[1][0]).abs().max()) < 1e-5 * scale and float((outs[0][1] - outs[1][1]).abs().max()) < 1e-5 * scale


def test_tiled_op_in_pieces_equals_the_one_call(tile_layout):
    """vican_tiled_op (operand at the descriptors' address) + vican_slab_reduce_fx per tile - the pieces a caller may still use -
    against vican_tiled_op_z (operand and result in the caller's arrays, one fold launch): the same bits."""
    if tile_layout != "fused":
        pytest.skip("the fused launch only")
    import ctypes as C
    from vican_amd import _lib
    from vican_amd.device import _ptr, _stream
    Cn, T, tile = 90, 300, 32
    rp, col, blk, a, w, u, v = random_graph(Cn, T, 3, 9, 5, False)
    g, K = _tiled(Cn, rp, col, blk, a, w, u, v, np.float64, tile)
    rng = np.random.default_rng(3)
    x = K.from_numpy(np.linalg.qr(rng.standard_normal((3 * Cn, 3)))[0])
    lam, cc = K.empty(T, 9), K.empty(Cn)
    K.init_duals(lam, cc)
    z1, z2 = K.empty(3 * Cn, 3), K.empty(3 * Cn, 3)
    K.block_op(lam, x, z1)
    f = K._fused
    assert f is not None
    f.x.copy_(x)
    _lib.check(K.lib.vican_tiled_op(C.cast(f.host, C.c_void_p), _ptr(f.dev), len(K.tiles), f.nwgt, _ptr(lam), f.parity, _stream()), "vican_tiled_op")
    f.parity ^= 1
    for k, Kt in enumerate(K.tiles):
        r0, r1 = K._tile_rows(k)
        fxp = Kt.g.fx.data_ptr()
        _lib.check(K.lib.vican_slab_reduce_fx(_ptr(f.zpart[k]), f.nwgt, Kt.C, 9, 1.0, C.c_void_p(fxp + 24), C.c_void_p(fxp + 56), _ptr(z2[r0:r1]),
                                              _stream()), "vican_slab_reduce_fx")
    assert torch.equal(z1, z2)
    zt = torch.empty(3, 3 * Cn, dtype=torch.float64, device=z1.device).t()           # a result array that is not contiguous
    K.block_op(lam, x.t().contiguous().t(), zt)
    assert torch.equal(zt, z1)
