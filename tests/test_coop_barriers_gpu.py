"""The cooperative kernels' grid barriers under contention (include/vican_hip.h: vican_set_barrier_abort): a filler kernel
(vican_test_occupy) keeps compute units busy on a second stream while the cooperative kernels run.

* filler shorter than the spin limit: part of the cooperative grid waits for a free compute unit, the resident part spins
  at its barrier - the launch completes late, bit-identical to the undisturbed one, nothing hangs;
* filler longer than the (shortened) spin limit: the resident workgroups raise the abort word and leave, the host sees it
  without any copy, re-arms, and the launch-sequence path delivers the result; the drop-in API does all of that by itself."""
import ctypes as C
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_kernels_gpu import make_backends       # noqa: E402


def occupy(lib, n_wg, usec, stream, lds=150 * 1024, threads=1024):
    from vican_amd import _lib
    _lib.check(lib.vican_test_occupy(n_wg, threads, lds, int(usec), C.c_void_p(stream.cuda_stream)), "vican_test_occupy")


def coop_step_setup(C_=1000, j=6):
    H, N, _ = make_backends(C_, 40, 1, 6, 9, np.float64)
    rng = np.random.default_rng(12)
    n, m = 3 * C_, 24
    Q, _ = np.linalg.qr(rng.standard_normal((n, 3 * (j + 1))))
    Vn = np.zeros((3 * (m + 1), n)); Vn[: 3 * (j + 1)] = Q.T
    lam = rng.standard_normal((C_, 3, 3)); lam = (lam @ np.swapaxes(lam, 1, 2) + np.eye(3)).reshape(C_, 9)
    lamd, zd, V0 = H.from_numpy(lam), H.from_numpy(rng.standard_normal((n, 3))), H.from_numpy(Vn.reshape(-1).copy())
    bufs = dict(R=H.zeros(3 * n), Hs=H.zeros(3 * (m + 1) * 3), G=H.zeros(9), Hcol=H.zeros(3 * (m + 1) * 3), beta=H.zeros(9), x=H.zeros(n, 3))

    def step(V):
        H.lanczos_cam_step(lamd, V, n, j, zd, bufs["R"], bufs["Hs"], bufs["G"], bufs["Hcol"], bufs["beta"], bufs["x"], 0.0)
        return [t.clone() for t in (V, bufs["Hcol"], bufs["beta"], bufs["x"])]
    return H, V0, step


def test_abort_word_is_host_visible_and_free_to_poll():
    from vican_amd.device import barrier_abort_word
    w = barrier_abort_word()
    assert w.is_pinned() and w.dtype == torch.int32 and int(w[0]) == 0


def test_cooperative_step_waits_for_a_busy_device_and_repeats_bit_identically():
    """All but a few compute units are held by the filler for 30 ms: the 32 workgroups of the cooperative step cannot all
    be resident at first (the resident ones spin), the launch ends when the filler does - same bits, no abort."""
    H, V0, step = coop_step_setup()
    assert H.coop_cam_step
    ref = step(V0.clone())
    side = torch.cuda.Stream()
    for free_cus in (0, 4, 12):
        torch.cuda.synchronize()
        occupy(H.lib, 256 - free_cus, 30_000, side)
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        out = step(V0.clone())
        t1.record()
        torch.cuda.synchronize()
        assert not H.barrier_aborted() and H.coop_cam_step
        for a, b in zip(out, ref):
            assert torch.equal(a, b)
        print("cooperative step behind a filler on %d compute units: %.1f ms" % (256 - free_cus, t0.elapsed_time(t1)))


def test_cooperative_step_aborts_instead_of_hanging_and_the_launch_sequence_takes_over():
    """Spin limit 3 ms, filler 300 ms on all but four compute units: the few resident workgroups give up, the host sees the
    abort word, and the same step through the launch sequence agrees with the undisturbed cooperative result to rounding."""
    from vican_amd.device import barrier_abort_word
    H, V0, step = coop_step_setup()
    ref = step(V0.clone())
    barrier_abort_word(timeout_us=3000)
    try:
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        occupy(H.lib, 252, 300_000, side)
        step(V0.clone())
        torch.cuda.current_stream().synchronize()
        if not H.barrier_aborted():
            pytest.skip("the whole cooperative grid found room next to the filler on this device")
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            H.cooperative_failed("test")
        assert caught and not H.coop_cam_step and not H.barrier_aborted()
        assert int(H._coop_sync.abs().sum()) == 0
        torch.cuda.synchronize()
        out = step(V0.clone())                              # launch-sequence path now
        for a, b in zip(out, ref):
            assert float((a - b).abs().max()) < 1e-10
    finally:
        barrier_abort_word(timeout_us=0)                    # back to the default limit
        torch.cuda.synchronize()


def test_resident_cg_falls_back_when_its_grid_is_kept_out():
    """vican_cg_resident behind a filler that outlasts the (shortened) spin limit: done = -1 comes back, the solver runs the
    multi-kernel path and returns the same solution (to rounding) as an undisturbed resident solve."""
    from vican_amd.device import barrier_abort_word
    from vican_amd.solver import Comm, TranslationSolver
    H, N, g = make_backends(100, 4000, 2, 9, 21, np.float64, "wave")
    assert H.cg_resident_ok
    eye = torch.eye(3, dtype=torch.float64, device="cuda")
    rc, rt = eye.repeat(100, 1, 1).reshape(-1, 3).contiguous(), eye.repeat(4000, 1, 1).reshape(4000, 9).contiguous()

    def solve():
        tr = TranslationSolver(H, Comm.single(), rtol=1e-9)
        tr.setup(rc, rt)
        xc, xt = tr.solve(3 * 4100)
        torch.cuda.synchronize()
        return xc.clone(), xt.clone(), tr.info
    xc0, xt0, info0 = solve()
    assert info0.get("resident") and info0["converged"]
    barrier_abort_word(timeout_us=3000)
    try:
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        occupy(H.lib, 252, 300_000, side)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            xc1, xt1, info1 = solve()
        if info1.get("resident"):
            pytest.skip("the resident grid found room next to the filler on this device")
        assert caught and info1["converged"] and not H.cg_resident_ok and not H.barrier_aborted()
        assert abs(info1["cg_iters"] - info0["cg_iters"]) <= 1
        assert float((xc1 - xc0).abs().max()) < 1e-7 and float((xt1 - xt0).abs().max()) < 1e-7
    finally:
        barrier_abort_word(timeout_us=0)
        torch.cuda.synchronize()


def test_dropin_solve_survives_a_busy_device():
    """End to end: the drop-in API while a filler holds most compute units for 0.3 s with a 3 ms spin limit - whatever the
    cooperative kernels do (wait, or abort and fall back), the poses are those of an undisturbed solve."""
    from vican_amd import synth
    from vican_amd.bipgo import bipartite_se3sync
    from vican_amd.device import barrier_abort_word
    from vican_amd.geometry import SE3, geodesic
    scene = synth.make_scene(n_cam=40, n_time=600, n_marker=6, seed=3)
    flat = synth.make_camera_edges(scene, cpt=3, mpv=2, sigma_r=1e-3, sigma_t=1e-3, seed=4)
    src, cons = synth.edges_to_dict(flat, SE3), synth.constraints_from_scene(scene, SE3)
    unit, keep = (lambda e: 1.0), (lambda e: True)
    ref = bipartite_se3sync(src, cons, unit, unit, keep, 4, "conjugate_gradient", np.float64)
    barrier_abort_word(timeout_us=3000)
    try:
        lib = __import__("vican_amd._lib", fromlist=["load"]).load()
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        occupy(lib, 252, 300_000, side)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = bipartite_se3sync(src, cons, unit, unit, keep, 4, "conjugate_gradient", np.float64)
    finally:
        barrier_abort_word(timeout_us=0)
        torch.cuda.synchronize()
    R = np.stack([out[k].R() for k in out]); Rr = np.stack([ref[k].R() for k in ref])
    t = np.stack([out[k].t() for k in out]); tr = np.stack([ref[k].t() for k in ref])
    assert float(geodesic(R, Rr).max()) < 1e-9 and float(np.linalg.norm(t - tr, axis=1).max()) < 1e-7


def _tiled_setup():
    from vican_amd.backend_cpu import NumpyBackend
    from test_kernels_gpu import random_graph
    from vican_amd.device import TiledBackend, TiledGraph
    C_, T, tile = 600, 4000, 300
    rp, col, blk, a, w, u, v = random_graph(C_, T, 40, 60, 8, False)
    dev = torch.device("cuda:0")
    to = lambda x: torch.from_numpy(x).to(dev)
    g = TiledGraph(C_, to(rp), to(col), to(blk), to(a), to(w), to(u), to(v), tile=tile)
    K = TiledBackend(g)
    N = NumpyBackend(C_, rp, col, blk, a, w, u, v, storage=np.float64)
    lh, ch, ln, cn = K.empty(T, 9), K.empty(C_), N.empty(T, 9), N.empty(C_)
    K.init_duals(lh, ch); N.init_duals(ln, cn)
    x = np.linalg.qr(np.random.default_rng(3).standard_normal((3 * C_, 3)))[0]
    zn = N.empty(3 * C_, 3)
    N.block_op(ln, N.from_numpy(x), zn)
    return K, lh, K.from_numpy(x), zn.numpy()


def test_fused_tiled_operator_waits_for_a_busy_device_and_repeats_bit_identically():
    """vican_tiled_op (one workgroup per compute unit, wavefronts that wait for other tiles' row-sum shares) behind a filler that
    holds most compute units for 30 ms: the resident workgroups spin on shares of workgroups that are not running yet, the
    launch ends when the filler does - same bits as the undisturbed launch, no abort."""
    K, lh, x, zn = _tiled_setup()
    assert K._fused is not None
    z0, z1 = K.empty(zn.shape[0], 3), K.empty(zn.shape[0], 3)
    K.block_op(lh, x, z0)
    torch.cuda.synchronize()
    assert np.abs(z0.cpu().numpy() - zn).max() <= 1e-10 * np.abs(zn).max()
    side = torch.cuda.Stream()
    for free_cus in (0, 16):
        torch.cuda.synchronize()
        occupy(K.lib, 256 - free_cus, 30_000, side)
        K.block_op(lh, x, z1)
        torch.cuda.synchronize()
        assert K._fused is not None and not K.barrier_aborted(), K.coop_failures
        assert torch.equal(z0, z1)


def test_fused_tiled_operator_aborts_instead_of_hanging_and_the_two_pass_path_takes_over():
    """Spin limit 3 ms, filler 300 ms on all but ONE compute unit (the workgroups that exchange shares are neighbours in
    dispatch order - blockIdx % n_tile is the tile -, so a few free compute units are enough for the launch to make progress
    pair by pair): the one resident workgroup gives up waiting for the shares of its partner that cannot start, the host sees
    the abort word, the backend drops the fused launch and the rows pass + camera pass per tile deliver the result."""
    from vican_amd.device import barrier_abort_word
    K, lh, x, zn = _tiled_setup()
    assert K._fused is not None
    z = K.empty(zn.shape[0], 3)
    barrier_abort_word(timeout_us=3000)
    try:
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        occupy(K.lib, 255, 300_000, side)
        K.block_op(lh, x, z)
        torch.cuda.current_stream().synchronize()
        if not K.barrier_aborted():
            pytest.skip("the fused grid found room next to the filler on this device")
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            K.cooperative_failed("test")
        assert caught and K._fused is None and not K.barrier_aborted()
        torch.cuda.synchronize()
        K.block_op(lh, x, z)                                 # two passes per tile now
        torch.cuda.synchronize()
        assert np.abs(z.cpu().numpy() - zn).max() <= 1e-10 * np.abs(zn).max()
    finally:
        barrier_abort_word(timeout_us=0)
        torch.cuda.synchronize()
