"""Edge cases of the hot path on the GPU: ragged / degenerate graphs, size limits, error behaviour."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from vican_amd.backend_cpu import NumpyBackend                                   # noqa: E402
from test_kernels_gpu import make_backends                                 # noqa: E402
from vican_amd import synth                                                # noqa: E402
from vican_amd._lib import VicanError                                      # noqa: E402
from vican_amd.solver import Comm, solve_on_backend                        # noqa: E402


def test_maximum_camera_count_fits_lds():
    """C = 1024 is the largest camera set of the LDS-resident sweep (f32 blocks); one more must fail loudly."""
    C, T = 1024, 48
    H, N, g = make_backends(C, T, 200, 1024, 77, np.float32, None, None, False)
    rng = np.random.default_rng(0)
    x = np.linalg.qr(rng.standard_normal((3 * C, 3)))[0]
    lamT_h, cd_h, lamT_n, cd_n = H.empty(T, 9), H.empty(C), N.empty(T, 9), N.empty(C)
    H.init_duals(lamT_h, cd_h); N.init_duals(lamT_n, cd_n)
    zh, zn = H.empty(3 * C, 3), N.empty(3 * C, 3)
    H.block_op(lamT_h, H.from_numpy(x), zh); N.block_op(lamT_n, N.from_numpy(x), zn)
    assert np.abs(zh.cpu().numpy() - zn.numpy()).max() <= 2e-6 * np.abs(zn.numpy()).max()
    from vican_amd.device import LocalGraph
    dev = torch.device("cuda:0")
    with pytest.raises(VicanError):
        LocalGraph(1025, torch.tensor([0, 1], dtype=torch.int32, device=dev), torch.tensor([3], dtype=torch.int32, device=dev),
                   torch.eye(3, device=dev).reshape(1, 9), torch.ones(1, device=dev))


def test_single_row_and_tiny_graph():
    """T = 1 row seen by 2 cameras: everything degenerates gracefully (one chunk, one workgroup)."""
    H, N, g = make_backends(2, 1, 2, 2, 5, np.float64, 256, None, False)
    assert g.n_chunk == 1 and g.n_wg == 1
    x = np.linalg.qr(np.random.default_rng(1).standard_normal((6, 3)))[0]
    lamT_h, cd_h, lamT_n, cd_n = H.empty(1, 9), H.empty(2), N.empty(1, 9), N.empty(2)
    H.init_duals(lamT_h, cd_h); N.init_duals(lamT_n, cd_n)
    zh, zn = H.empty(6, 3), N.empty(6, 3)
    H.block_op(lamT_h, H.from_numpy(x), zh); N.block_op(lamT_n, N.from_numpy(x), zn)
    assert np.abs(zh.cpu().numpy() - zn.numpy()).max() < 1e-11


def test_degree_one_timesteps_are_legal():
    """Timesteps seen by a single camera only add a diagonal block that cancels in L;
    the solve must still agree with the NumPy restatement."""
    from vican_amd.device import HipBackend, LocalGraph
    C, T = 6, 200
    dev = torch.device("cuda:0")
    gr = synth.make_merged_graph_torch(C, T, 3, dev, torch.float64, seed=9, sigma_r=1e-3, sigma_t=1e-3)
    # drop all but the first edge of every third row -> degree-1 rows
    rp = gr["row_ptr"].cpu().numpy().astype(np.int64)
    keep = np.ones(int(rp[-1]), dtype=bool)
    for t in range(0, T, 3):
        keep[rp[t] + 1: rp[t + 1]] = False
    deg = np.array([keep[rp[t]:rp[t + 1]].sum() for t in range(T)])
    rp2 = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    kt = torch.from_numpy(keep).to(dev)
    arrs = {k: gr[k][kt] for k in ("col", "blk", "a", "w", "u", "v")}
    g = LocalGraph(C, torch.from_numpy(rp2).to(dev), arrs["col"], arrs["blk"], arrs["a"], arrs["w"], arrs["u"], arrs["v"])
    K = HipBackend(g)
    Nn = NumpyBackend(C, rp2, arrs["col"].cpu().numpy(), arrs["blk"].cpu().numpy(), arrs["a"].cpu().numpy(),
                      arrs["w"].cpu().numpy(), arrs["u"].cpu().numpy(), arrs["v"].cpu().numpy())
    out_h = solve_on_backend(K, Comm(), 4, 3 * (C + T))
    out_n = solve_on_backend(Nn, Comm(), 4, 3 * (C + T))
    assert np.abs(out_h[0].cpu().numpy() - out_n[0].numpy()).max() < 1e-8          # camera rotations
    assert np.abs(out_h[1].cpu().numpy()[:T] - out_n[1].numpy()[:T]).max() < 1e-7  # timestep rotations
    assert abs(out_h[4]["cg_iters"] - out_n[4]["cg_iters"]) <= 1


def test_maxiter_zero_raises_like_the_reference():
    """maxiter = 0: the reference's loop body never runs and it dies on the unbound `r_c` (bipgo.py:346); the drop-in
    must not hand uninitialised device buffers to the translation stage."""
    from vican_amd.bipgo import bipartite_se3sync, object_bipartite_se3sync
    from vican_amd.geometry import SE3
    scene = synth.make_scene(n_cam=4, n_time=10, n_marker=3, seed=1)
    src = synth.edges_to_dict(synth.make_camera_edges(scene, cpt=2, mpv=2, seed=2), SE3)
    cons = synth.constraints_from_scene(scene, SE3)
    one = lambda e: 1.0
    with pytest.raises(UnboundLocalError, match="r_c"):
        bipartite_se3sync(src, cons, one, one, lambda e: True, 0, "conjugate_gradient", np.float64)
    osrc = synth.edges_to_dict(synth.make_object_edges(scene, mpv=3, seed=3), SE3)
    with pytest.raises(UnboundLocalError, match="r_c"):
        object_bipartite_se3sync(osrc, one, one, lambda e: True, 0, "conjugate_gradient", np.float32)


def test_api_error_paths():
    from vican_amd.bipgo import bipartite_se3sync
    from vican_amd.geometry import SE3
    scene = synth.make_scene(n_cam=4, n_time=10, n_marker=3, seed=1)
    flat = synth.make_camera_edges(scene, cpt=2, mpv=2, seed=2)
    src = synth.edges_to_dict(flat, SE3)
    cons = synth.constraints_from_scene(scene, SE3)
    one = lambda e: 1.0
    with pytest.raises(ValueError):                               # filter rejects everything
        bipartite_se3sync(src, cons, one, one, lambda e: False, 4, "conjugate_gradient", np.float64)
    out = bipartite_se3sync(src, cons, one, one, lambda e: True, 1, "direct", np.float32)     # maxiter 1, LSQR, f32
    assert len(out) == 4 + 10 and next(iter(out.values())).R().dtype == np.float32


def test_disconnected_graph_warns_and_still_returns():
    """Two unrelated rigs in one edge dict: the reference returns an arbitrary null-space mixture silently
    (SURVEY.md section 5); the drop-in warns, solves, and each component is still internally consistent."""
    import golden_cases as gc
    from util import load_golden, rebuild_inputs
    from vican.bipgo import bipartite_se3sync
    from vican_amd.bipgo import DisconnectedGraphWarning
    from vican_amd.geometry import geodesic
    g = load_golden("g2_small")
    case, src, cons, (nr, nt, ff) = rebuild_inputs("g2_small", g)
    both = dict(src)
    both.update({("x" + c, "9" + tm): v for (c, tm), v in src.items()})
    info = {}
    with pytest.warns(DisconnectedGraphWarning, match="2 connected components"):
        res = bipartite_se3sync(both, cons, nr, nt, ff, gc.MAXITER, "conjugate_gradient", np.float64, info=info)
    assert len(res) == 2 * (len(g["gt_R_cam"]) + len(g["gt_R_obj"]))
    assert all(np.isfinite(p.R()).all() and np.isfinite(p.t()).all() for p in res.values())
    # inside the component of the gauge camera relative rotations equal those of the single-rig solve
    # (two components = six null vectors: all five eigenvalues the reference would see are ~0, so its loop ends after
    #  the first iteration - bipgo.py:283 - and so does the product's when its fourth and fifth Ritz values say so)
    n_it = info["early_exit"] or gc.MAXITER
    assert len(info["lanczos_steps"]) == n_it
    one = bipartite_se3sync(src, cons, nr, nt, ff, n_it, "conjugate_gradient", np.float64)
    keys = [k for k in one if "_" not in k]
    ra = np.stack([res[k].R() @ res[keys[0]].R().T for k in keys])
    rb = np.stack([one[k].R() @ one[keys[0]].R().T for k in keys])
    assert float(geodesic(ra, rb).max()) < 1e-6
