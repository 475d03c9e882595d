"""The four-call boundary of SURVEY.md 8(b) (include/vican_hip.h: vican_plan_create / vican_solve_rot / vican_solve_trans /
vican_plan_destroy, csrc/vican_facade.hip) driven through ``ctypes`` alone - no vican_amd.solver, no vican_amd.device: what a
maintainer of the reference would bind.  Inputs: the merged CSR problem of a golden case (host front-end), uploaded with torch
as the allocator; outputs against the REAL reference's poses (tests/golden) with the tolerances of the drop-in tests."""
import ctypes as C

import numpy as np
import pytest
import torch

import golden_cases as gc
from util import e2e_translation_tol, expected, iteration_slack, load_golden, rebuild_inputs
from vican_amd import _lib, frontend
from vican_amd.geometry import geodesic

pytestmark = pytest.mark.gpu


def solve_through_the_facade(prob, dt, maxiter=gc.MAXITER, comm=None):
    """comm: a vican_comm_t* - the plan then is ONE RANK of a sharded solve (vican_plan_set_comm)."""
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    tdt = torch.float32 if dt == "float32" else torch.float64
    up = lambda a, d: torch.from_numpy(np.ascontiguousarray(a)).to(dev, d).contiguous()
    row_ptr, col = up(prob.row_ptr, torch.int32), up(prob.col, torch.int32)
    blk, a = up(prob.blk, tdt), up(prob.a, tdt)
    w, u, v = up(prob.w, torch.float64), up(prob.u, torch.float64), up(prob.v, torch.float64)
    deg_t, deg_c = up(prob.deg_t, torch.float64), up(prob.deg_c, torch.float64)
    Cn, T, E = prob.n_cam, prob.n_time, prob.n_edges
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    plan = C.c_void_p()
    rc = lib.vican_plan_create(Cn, T, E, _lib.STORE_F32 if dt == "float32" else _lib.STORE_F64, p(row_ptr), p(col), p(blk), p(a), p(w), p(u), p(v),
                               p(deg_t), p(deg_c), stream, C.byref(plan))
    assert rc == 0, lib.vican_last_error()
    try:
        g = _lib.Graph()
        assert lib.vican_plan_describe(plan, C.byref(g)) == 0
        assert g.n_cam == Cn and g.n_time == T and g.n_chunk >= 1 and g.n_wg >= 1
        if comm is not None:
            assert lib.vican_plan_set_comm(plan, comm, stream) == 0, lib.vican_last_error()
        rcs, Rt = torch.empty(3 * Cn, 3, dtype=torch.float64, device=dev), torch.empty(T, 9, dtype=torch.float64, device=dev)
        x_c, x_t = torch.empty(Cn, 3, dtype=torch.float64, device=dev), torch.empty(T, 3, dtype=torch.float64, device=dev)
        info = _lib.SolveInfo()
        rc = lib.vican_solve_rot(plan, maxiter, 1e-10, p(rcs), p(Rt), C.byref(info), stream)
        assert rc == 0, lib.vican_last_error()
        assert info.iterations == maxiter and info.lanczos_steps >= maxiter
        rc = lib.vican_solve_trans(plan, p(rcs), p(Rt), 1e-5, 0, p(x_c), p(x_t), C.byref(info), stream)
        assert rc == 0, lib.vican_last_error()
        assert info.cg_converged == 1
    finally:
        assert lib.vican_plan_destroy(plan) == 0
    Rc = np.swapaxes(rcs.cpu().numpy().reshape(Cn, 3, 3), 1, 2)          # world<-node, as the reference returns them (bipgo.py:346)
    Rtt = np.swapaxes(Rt.cpu().numpy().reshape(T, 3, 3), 1, 2)
    return Rc, Rtt, x_c.cpu().numpy(), x_t.cpu().numpy(), info, g


@pytest.mark.parametrize("name,dt", [("g2_small", "float64"), ("g2_small", "float32"), ("g3_medium", "float64"), ("g3_medium", "float32"),
                                     ("g5_strings", "float64")])
def test_four_calls_reproduce_the_reference(name, dt):
    g = load_golden(name)
    case, src, cons, (nr, nt, ff) = rebuild_inputs(name, g)
    exp = expected(g, "conjugate_gradient", dt)
    prob = frontend.flatten(src, cons, nr, nt, ff, np.dtype(dt).type)
    Rc, Rt, pc, pt, info, graph = solve_through_the_facade(prob, dt)
    rot, pos = {}, {}
    for i, c in enumerate(prob.cam_names):
        rot[str(c)], pos[str(c)] = Rc[i], pc[i]
    for i, s in enumerate(prob.time_names):
        rot[str(s) + "_0"], pos[str(s) + "_0"] = Rt[i], pt[i]
    keys = [str(k) for k in exp["keys"]]
    R = np.stack([rot[k] for k in keys]); t = np.stack([pos[k] for k in keys])
    r_err, t_err = float(geodesic(R, exp["R"]).max()), float(np.linalg.norm(t - exp["t"], axis=1).max())
    print("%s %s through the four calls: rot %.2e rad, trans %.2e m, %d Lanczos steps, %d sweeps, cg %d vs %d, layout %s" % (
        name, dt, r_err, t_err, info.lanczos_steps, info.sweeps, info.cg_iters, int(exp["cg_iters"]), "wave" if graph.layout == 1 else "block"))
    from conftest import record_parity
    record_parity(name, dt, "facade", r_err, t_err, e2e_translation_tol(name, dt), info.cg_iters, int(exp["cg_iters"]))
    assert r_err < (5e-6 if dt == "float32" else 1e-7), r_err
    assert t_err < e2e_translation_tol(name, dt), t_err
    assert abs(info.cg_iters - int(exp["cg_iters"])) <= iteration_slack(name, dt)
    evr = np.sort(exp["evals"], axis=1)[-1, :3]
    assert np.abs(np.sort(np.array(info.evals[:3])) - evr).max() < (1e-4 if dt == "float32" else 1e-7) * np.abs(exp["evals"]).max()


@pytest.mark.parametrize("name,dt", [("g3_medium", "float64"), ("g3_medium", "float32"), ("g2_small", "float64")])
def test_the_sharded_schedule_through_the_four_calls(name, dt):
    """vican_plan_set_comm: the plan as one rank of a timestep-sharded solve - here the ONLY rank, with a communicator whose
    all-reduces are launches of the peer exchange (vican_comm_create_local + vican_comm_peer_export / _attach, ctypes alone: the
    rank's own mailbox slot is its peer).  Every sweep's camera partial and both messages of every CG iteration go through the
    exchange; poses against the real reference's, and against the single-rank facade solve of the same plan inputs."""
    lib = _lib.load()
    g = load_golden(name)
    case, src, cons, (nr, nt, ff) = rebuild_inputs(name, g)
    exp = expected(g, "conjugate_gradient", dt)
    prob = frontend.flatten(src, cons, nr, nt, ff, np.dtype(dt).type)
    comm = C.c_void_p()
    assert lib.vican_comm_create_local(0, 1, C.byref(comm)) == 0
    assert lib.vican_comm_peer_export(comm, 9 * 1024 + 96, C.create_string_buffer(64)) == 0, lib.vican_last_error()
    assert lib.vican_comm_peer_attach(comm, None) == 0, lib.vican_last_error()
    try:
        Rc, Rt, pc, pt, info, graph = solve_through_the_facade(prob, dt, comm=comm)
        assert lib.vican_comm_peer_status(comm) == 0
    finally:
        lib.vican_comm_destroy(comm)
    Rc1, Rt1, pc1, pt1, info1, _ = solve_through_the_facade(prob, dt)
    rot, pos = {}, {}
    for i, c in enumerate(prob.cam_names):
        rot[str(c)], pos[str(c)] = Rc[i], pc[i]
    for i, s_ in enumerate(prob.time_names):
        rot[str(s_) + "_0"], pos[str(s_) + "_0"] = Rt[i], pt[i]
    keys = [str(k) for k in exp["keys"]]
    R = np.stack([rot[k] for k in keys]); t = np.stack([pos[k] for k in keys])
    r_err, t_err = float(geodesic(R, exp["R"]).max()), float(np.linalg.norm(t - exp["t"], axis=1).max())
    from conftest import record_parity
    record_parity(name, dt, "facade, sharded schedule", r_err, t_err, e2e_translation_tol(name, dt), info.cg_iters, int(exp["cg_iters"]))
    assert r_err < (5e-6 if dt == "float32" else 1e-7), r_err
    assert t_err < e2e_translation_tol(name, dt), t_err
    assert abs(info.cg_iters - int(exp["cg_iters"])) <= iteration_slack(name, dt)
    # against the single-rank schedule of the same boundary: the same answer to the eigen-solver's tolerance
    assert float(geodesic(Rc, Rc1).max()) < (2e-6 if dt == "float32" else 1e-9) and abs(info.cg_iters - info1.cg_iters) <= 1


def _poses_against_reference(prob, exp, Rc, Rt, pc, pt):
    rot, pos = {}, {}
    for i, c in enumerate(prob.cam_names):
        rot[str(c)], pos[str(c)] = Rc[i], pc[i]
    for i, s_ in enumerate(prob.time_names):
        rot[str(s_) + "_0"], pos[str(s_) + "_0"] = Rt[i], pt[i]
    keys = [str(k) for k in exp["keys"]]
    R = np.stack([rot[k] for k in keys]); t = np.stack([pos[k] for k in keys])
    return float(geodesic(R, exp["R"]).max()), float(np.linalg.norm(t - exp["t"], axis=1).max())


@pytest.fixture
def tile_cams():
    """vican_facade_set_tile_cams (include/vican_hip_test.h): plans made inside the test cut their cameras into tiles this wide."""
    lib = _lib.load()

    def set_(n):
        assert lib.vican_facade_set_tile_cams(n) == 0, lib.vican_last_error()
    yield set_
    lib.vican_facade_set_tile_cams(1024)


@pytest.mark.parametrize("name,dt,tile", [("g3_medium", "float64", 16), ("g3_medium", "float32", 16), ("g2_small", "float64", 3),
                                          ("g2_small", "float32", 3), ("g9_large_shop", "float32", 128),
                                          # more than four tiles: the CG product tile by tile, the operator's share reload loop
                                          ("g3_medium", "float64", 6), ("g3_medium", "float32", 5)])
@pytest.mark.parametrize("sharded", [False, True])
def test_camera_tiles_behind_the_four_calls(name, dt, tile, sharded, tile_cams):
    """More cameras than a tile holds (csrc/vican_facade_tiles.hip; forced here on golden cases: 40 cameras in tiles of 16, 6 in
    tiles of 3, large_shop's 340 in tiles of 128; 40 in tiles of 6 / 5 = seven / eight tiles): the same four calls, poses against the REAL reference's.  sharded: the tiled
    plan as the only rank of a sharded solve - its all-reduces (operator result, both CG messages) through the peer exchange."""
    lib = _lib.load()
    g = load_golden(name)
    if name == "g9_large_shop":             # (inputs regenerated from their seed, as tests/test_large_shop_scale.py does)
        from vican_amd import synth
        from vican_amd.geometry import SE3
        scene, flat = gc.build_flat(gc.LARGE_SHOP)
        src, cons = synth.edges_to_dict(flat, SE3), synth.constraints_from_scene(scene, SE3)
        nr, nt, ff = (gc.CALLABLES[gc.LARGE_SHOP[k]] for k in ("noise_r", "noise_t", "filt"))
    else:
        case, src, cons, (nr, nt, ff) = rebuild_inputs(name, g)
    exp = expected(g, "conjugate_gradient", dt)
    prob = frontend.flatten(src, cons, nr, nt, ff, np.dtype(dt).type)
    tile_cams(tile)
    comm = None
    if sharded:
        comm = C.c_void_p()
        assert lib.vican_comm_create_local(0, 1, C.byref(comm)) == 0
        assert lib.vican_comm_peer_export(comm, 9 * 1024 + 96, C.create_string_buffer(64)) == 0, lib.vican_last_error()
        assert lib.vican_comm_peer_attach(comm, None) == 0, lib.vican_last_error()
    try:
        Rc, Rt, pc, pt, info, graph = solve_through_the_facade(prob, dt, comm=comm)
        if sharded:
            assert lib.vican_comm_peer_status(comm) == 0
    finally:
        if sharded:
            lib.vican_comm_destroy(comm)
    assert graph.n_cam == prob.n_cam and graph.n_wg >= 2             # (describe: the shared chunking, all tiles' workgroups)
    r_err, t_err = _poses_against_reference(prob, exp, Rc, Rt, pc, pt)
    print("%s %s through the four calls on tiles of %d cameras%s: rot %.2e rad, trans %.2e m, %d Lanczos steps, cg %d vs %d" % (
        name, dt, tile, " (sharded schedule)" if sharded else "", r_err, t_err, info.lanczos_steps, info.cg_iters, int(exp["cg_iters"])))
    from conftest import record_parity
    record_parity(name, dt, "facade, camera tiles" + (", sharded" if sharded else ""), r_err, t_err, e2e_translation_tol(name, dt),
                  info.cg_iters, int(exp["cg_iters"]))
    assert r_err < (5e-6 if dt == "float32" else 1e-7), r_err
    assert t_err < (2e-3 if name == "g9_large_shop" else e2e_translation_tol(name, dt)), t_err
    assert abs(info.cg_iters - int(exp["cg_iters"])) <= iteration_slack(name, dt, **({"extra": 2} if name == "g9_large_shop" else {}))


def test_1500_cameras_behind_the_four_calls_match_the_python_driver():
    """A graph that NEEDS tiles (1500 cameras: 2 x 752), synthetic, both storage types: vican_plan_create at its default tile width
    against the Python driver's tiled solve (vican_amd.tiled through bipgo.solve_problem's backend choice) of the same arrays."""
    from vican_amd import synth
    from vican_amd.solver import Comm, RotationSolver, TranslationSolver
    from vican_amd.tiled import TiledBackend, TiledGraph
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    Cn, T, cpt = 1500, 3000, 60
    for tdt in (torch.float32, torch.float64):
        gr = synth.make_merged_graph_torch(Cn, T, cpt, dev, tdt, seed=11)
        E = int(gr["col"].numel())
        p = lambda t: C.c_void_p(t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        plan = C.c_void_p()
        rc = lib.vican_plan_create(Cn, T, E, _lib.STORE_F32 if tdt == torch.float32 else _lib.STORE_F64, p(gr["row_ptr"]), p(gr["col"]), p(gr["blk"]),
                                   p(gr["a"]), p(gr["w"]), p(gr["u"]), p(gr["v"]), None, None, stream, C.byref(plan))
        assert rc == 0, lib.vican_last_error()
        try:
            rcs, Rt = torch.empty(3 * Cn, 3, dtype=torch.float64, device=dev), torch.empty(T, 9, dtype=torch.float64, device=dev)
            x_c, x_t = torch.empty(Cn, 3, dtype=torch.float64, device=dev), torch.empty(T, 3, dtype=torch.float64, device=dev)
            info = _lib.SolveInfo()
            assert lib.vican_solve_rot(plan, 4, 1e-10, p(rcs), p(Rt), C.byref(info), stream) == 0, lib.vican_last_error()
            assert lib.vican_solve_trans(plan, p(rcs), p(Rt), 1e-5, 0, p(x_c), p(x_t), C.byref(info), stream) == 0, lib.vican_last_error()
            assert info.cg_converged == 1 and float(rcs.abs().max()) > 0.5
            # LSQR on the same tiled plan (lsqr_solver="direct"): the least-squares solution CG's normal equations share - both stop
            # on loose tolerances, so they agree to those, not to the bit
            xl_c, xl_t = torch.empty_like(x_c), torch.empty_like(x_t)
            linfo = _lib.LsqrInfo()
            assert lib.vican_solve_trans_lsqr(plan, p(rcs), p(Rt), 0.0, 1e-9, 1e-9, 1e8, 0, p(xl_c), p(xl_t), C.byref(linfo), stream) == 0, lib.vican_last_error()
            assert linfo.istop in (1, 2) and linfo.itn > 3
            gauge = (xl_c - x_c).mean(0)                                    # (the system is singular: solutions differ by a translation of all nodes)
            d_l = max(float((xl_c - x_c - gauge).abs().max()), float((xl_t - x_t - gauge).abs().max()))
            print("1500 cameras, %s: LSQR against CG on the tiled plan: %.2e (scale %.2e), itn %d" % (str(tdt), d_l, float(x_c.abs().max()), linfo.itn))
            assert d_l < 2e-3 * float(x_c.abs().max())
        finally:
            assert lib.vican_plan_destroy(plan) == 0
        # a rotation-only plan of the same graph (no translation arrays): the same rotations to the bit, translations refused
        plan = C.c_void_p()
        assert lib.vican_plan_create(Cn, T, E, _lib.STORE_F32 if tdt == torch.float32 else _lib.STORE_F64, p(gr["row_ptr"]), p(gr["col"]), p(gr["blk"]),
                                     p(gr["a"]), None, None, None, None, None, stream, C.byref(plan)) == 0, lib.vican_last_error()
        try:
            rc2, Rt2 = torch.empty_like(rcs), torch.empty_like(Rt)
            assert lib.vican_solve_rot(plan, 4, 1e-10, p(rc2), p(Rt2), None, stream) == 0, lib.vican_last_error()
            assert torch.equal(rc2, rcs) and torch.equal(Rt2, Rt)
            assert lib.vican_solve_trans(plan, p(rc2), p(Rt2), 1e-5, 0, p(x_c.clone()), p(x_t.clone()), None, stream) == _lib.ERR_ARG
        finally:
            assert lib.vican_plan_destroy(plan) == 0
        # (rows packed for the shared chunking on both sides: vican_plan_rows_multi - the plan and the TiledGraph keep their rows in
        #  the same order of their own and hand per-row results back in the caller's)
        G = TiledGraph(Cn, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"], permute_rows=True)
        assert G.row_perm is not None
        K = TiledBackend(G)
        rot = RotationSolver(K, Comm(), eig_tol=1e-10)
        rc_ref, Rt_ref = rot.run(4)
        tr = TranslationSolver(K, Comm())
        tr.setup(rc_ref, Rt_ref)
        xc_ref, xt_ref = tr.solve(3 * (Cn + T))
        Rt_ref, xt_ref = G.unpermute_rows(Rt_ref.reshape(-1, 9)[:T]), G.unpermute_rows(xt_ref.reshape(-1, 3)[:T])
        tol_r = 2e-5 if tdt == torch.float32 else 1e-8
        d_rc = float((rcs - rc_ref.reshape(3 * Cn, 3)).abs().max()); d_rt = float((Rt - Rt_ref.reshape(T, 9)).abs().max())
        d_x = float((x_c - xc_ref.reshape(Cn, 3)).abs().max()); d_t = float((x_t - xt_ref[:T].reshape(T, 3)).abs().max())
        print("1500 cameras, %s: facade against the Python driver: rc %.2e Rt %.2e x_c %.2e x_t %.2e, cg %d vs %d" % (
            str(tdt), d_rc, d_rt, d_x, d_t, info.cg_iters, tr.info["cg_iters"]))
        assert d_rc < tol_r and d_rt < tol_r, (d_rc, d_rt)
        scale = float(xc_ref.abs().max())
        assert d_x < 1e-4 * scale and d_t < 1e-4 * scale, (d_x, d_t, scale)
        assert abs(info.cg_iters - tr.info["cg_iters"]) <= 1
        del K, rot, tr, gr


def test_tile_layouts_the_boundary_does_not_plan_are_refused_with_the_reason(tile_cams):
    """A tile without edges (cameras 8..15 never seen) is a layout only the host driver plans: VICAN_ERR_CAPACITY and a message."""
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    rng = np.random.default_rng(0)
    Cn, T = 16, 50
    cols = np.stack([np.sort(rng.choice(8, 3, replace=False)) for _ in range(T)]).reshape(-1).astype(np.int32)
    rp = (np.arange(T + 1) * 3).astype(np.int32)
    E = len(cols)
    up = lambda a, d: torch.from_numpy(np.ascontiguousarray(a)).to(dev, d)
    blk = up(np.tile(np.eye(3).reshape(9), (E, 1)), torch.float64); a = up(np.ones(E), torch.float64)
    p = lambda t: C.c_void_p(t.data_ptr())
    tile_cams(8)
    plan = C.c_void_p()
    rp_d, col_d = up(rp, torch.int32), up(cols, torch.int32)
    rc = lib.vican_plan_create(Cn, T, E, _lib.STORE_F64, p(rp_d), p(col_d), p(blk), p(a), None, None, None, None, None,
                               C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(plan))
    assert rc == _lib.ERR_CAPACITY and b"has no edges" in lib.vican_last_error(), lib.vican_last_error()
    assert not plan.value


def test_bad_arguments_are_refused():
    lib = _lib.load()
    plan = C.c_void_p()
    assert lib.vican_plan_create(0, 1, 1, 0, None, None, None, None, None, None, None, None, None, None, C.byref(plan)) == _lib.ERR_ARG
    assert lib.vican_solve_rot(None, 4, 1e-10, None, None, None, None) == _lib.ERR_ARG
    assert lib.vican_plan_set_comm(None, None, None) == _lib.ERR_ARG
    assert lib.vican_plan_destroy(None) == 0


@pytest.mark.parametrize("name,dt", [("g2_small", "float64"), ("g2_small", "float32"), ("g3_medium", "float64"), ("g5_strings", "float64")])
@pytest.mark.parametrize("tile", [0, 3])
def test_lsqr_through_the_facade_reproduces_the_reference(name, dt, tile, tile_cams):
    """vican_solve_trans_lsqr (lsqr_solver="direct", bipgo.py:479-480) behind the plan handle, ctypes alone: translations against
    the REAL reference's LSQR run with the tolerance of the drop-in test, scipy's istop and iteration count."""
    g = load_golden(name)
    exp = expected(g, "direct", dt)
    if not exp:
        pytest.skip("golden has no direct/%s run" % dt)
    case, src, cons, (nr, nt, ff) = rebuild_inputs(name, g)
    prob = frontend.flatten(src, cons, nr, nt, ff, np.dtype(dt).type)
    lib = _lib.load()
    if tile:
        tile_cams(tile)                     # (the same call on camera tiles: every pass over the edges tile by tile)
    dev = torch.device("cuda", torch.cuda.current_device())
    tdt = torch.float32 if dt == "float32" else torch.float64
    up = lambda a, d: torch.from_numpy(np.ascontiguousarray(a)).to(dev, d).contiguous()
    row_ptr, col = up(prob.row_ptr, torch.int32), up(prob.col, torch.int32)
    blk, a = up(prob.blk, tdt), up(prob.a, tdt)
    w, u, v = up(prob.w, torch.float64), up(prob.u, torch.float64), up(prob.v, torch.float64)
    Cn, T, E = prob.n_cam, prob.n_time, prob.n_edges
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    plan = C.c_void_p()
    assert lib.vican_plan_create(Cn, T, E, _lib.STORE_F32 if dt == "float32" else _lib.STORE_F64, p(row_ptr), p(col), p(blk), p(a), p(w), p(u), p(v),
                                 None, None, stream, C.byref(plan)) == 0, lib.vican_last_error()
    try:
        rcs, Rt = torch.empty(3 * Cn, 3, dtype=torch.float64, device=dev), torch.empty(T, 9, dtype=torch.float64, device=dev)
        x_c, x_t = torch.empty(Cn, 3, dtype=torch.float64, device=dev), torch.empty(T, 3, dtype=torch.float64, device=dev)
        assert lib.vican_solve_rot(plan, gc.MAXITER, 1e-10, p(rcs), p(Rt), None, stream) == 0, lib.vican_last_error()
        Rc_h = np.swapaxes(rcs.cpu().numpy().reshape(Cn, 3, 3), 1, 2)
        Rt_h = np.swapaxes(Rt.cpu().numpy().reshape(T, 3, 3), 1, 2)
        info = _lib.LsqrInfo()
        rc = lib.vican_solve_trans_lsqr(plan, p(rcs), p(Rt), frontend.bnorm2(prob, Rc_h, Rt_h), 1e-6, 1e-6, 1e8, 0, p(x_c), p(x_t), C.byref(info), stream)
        assert rc == 0, lib.vican_last_error()
        # a second solve on the same plan (workspace reused) gives the same bits
        x_c2, x_t2 = torch.empty_like(x_c), torch.empty_like(x_t)
        assert lib.vican_solve_trans_lsqr(plan, p(rcs), p(Rt), frontend.bnorm2(prob, Rc_h, Rt_h), 1e-6, 1e-6, 1e8, 0, p(x_c2), p(x_t2), None, stream) == 0
        torch.cuda.synchronize()
        assert torch.equal(x_c, x_c2) and torch.equal(x_t, x_t2)
    finally:
        assert lib.vican_plan_destroy(plan) == 0
    pos = {str(c): x_c.cpu().numpy()[i] for i, c in enumerate(prob.cam_names)}
    pos.update({str(s) + "_0": x_t.cpu().numpy()[i] for i, s in enumerate(prob.time_names)})
    t = np.stack([pos[str(k)] for k in exp["keys"]])
    t_err = float(np.linalg.norm(t - exp["t"], axis=1).max())
    print("%s %s LSQR through the facade: trans %.2e m, itn %d, istop %d" % (name, dt, t_err, info.itn, info.istop))
    assert t_err < (2e-6 if dt == "float64" else 5e-4), t_err
    assert info.istop in (1, 2)
    if "lsqr_iters" in exp:
        assert abs(info.itn - int(exp["lsqr_iters"])) <= 2
