"""Host logic added in round 3, on CPU: string ids -> sorted node indices without a comparison sort, the explicit
(machine-independent) arithmetic of the host merge against the reference-shaped formulation, scipy's J^T J entries, the
cooperative-kernel fallback wrapper, and the self-movement recorder the parity bounds are built on."""
import numpy as np
import pytest

import golden_cases as gc
from util import SelfMovement, load_golden, rebuild_inputs
from vican_amd import frontend, synth
from vican_amd.geometry import SE3
from vican_amd.solver import with_cooperative_fallback


@pytest.mark.parametrize("seed", range(6))
def test_sorted_codes_is_np_unique(seed):
    rng = np.random.default_rng(seed)
    alphabet = np.array(list("0123456789abcXYZ_-"))
    n = int(rng.integers(1, 400))
    ids = ["".join(rng.choice(alphabet, int(rng.integers(0 if seed else 1, 9)))) for _ in range(n)]
    if seed == 4:
        ids += ["é1", "10", "9"]                      # non-ASCII: falls back to np.unique
    if seed == 5:
        ids += ["123456789", "12345678"]              # longer than 8 characters: falls back
    for prefix in ("", "c"):
        names, idx = frontend.sorted_codes(ids, prefix)
        ref_names, ref_idx = np.unique(np.char.add(prefix, np.array(ids, dtype=str)), return_inverse=True)
        assert list(names) == list(ref_names) and np.array_equal(idx, ref_idx)


def test_node_order_is_the_string_order_of_the_reference():
    """bipgo.py:225-229 sorts 'c' + id / 't' + timestamp as STRINGS ('t10' < 't2'); the translation nodes are one mixed
    string-sorted list of camera ids and '<t>_0' (bipgo.py:426-430)."""
    from vican_amd.geometry import SE3
    ix = frontend.index_edges(["2", "10", "100", "2"], ["2", "10", "2", "10"], ["0", "0", "1", "1"],
                              {"0": SE3(pose=np.eye(4)), "1": SE3(pose=np.eye(4))})
    assert list(ix.cam_names) == ["10", "100", "2"] and list(ix.time_names) == ["10", "2"]
    assert list(ix.tnodes) == sorted(["10", "100", "2", "10_0", "2_0"])
    assert [ix.tnodes[i] for i in ix.tnode_of_cam] == ["10", "100", "2"] and [ix.tnodes[i] for i in ix.tnode_of_time] == ["10_0", "2_0"]
    assert ix.root == "0" and list(ix.ci) == [2, 0, 1, 2] and list(ix.ti) == [1, 0, 1, 0]


@pytest.mark.parametrize("name", ["g3_medium", "g4_illcond"])
def test_host_merge_is_sequential_and_matches_a_per_edge_loop(name):
    """merge_host against the reference-shaped per-edge loop (bipgo.py:203-221, 445-469 restated edge by edge): identical
    blocks up to the association (k R) (R_m^T R_root) vs ((k R) R_m^T) R_root, and - the part scipy fixes - J^T J entries that
    are float32 products summed in float32 in source order for dtype=float32."""
    case, src, cons, (nr, nt, ff) = rebuild_inputs(name)
    for dt in (np.float32, np.float64):
        prob = frontend.flatten(src, cons, nr, nt, ff, dt)
        root = str(min(cons.keys()))
        C, T = prob.n_cam, prob.n_time
        cpos = {c: i for i, c in enumerate(prob.cam_names)}; tpos = {t: i for i, t in enumerate(prob.time_names)}
        M, w = {}, {}
        dc, dtt = np.zeros(C, dtype=dt), np.zeros(T, dtype=dt)
        for key, val in src.items():
            if not ff(val):
                continue
            ts, mid = key[1].split("_")
            c, t = cpos[key[0]], tpos[ts]
            blk = nr(val) * np.asarray(val["pose"].R(), dtype=np.float64) @ np.asarray(cons[mid].R()).T @ np.asarray(cons[root].R())
            M[(t, c)] = M.get((t, c), 0.0) + blk
            k = dt(nt(val))
            w[(t, c)] = dt(w.get((t, c), dt(0)) + k * k)
            dc[c] = dc[c] + k * k; dtt[t] = dtt[t] + k * k
        rows = np.repeat(np.arange(T), np.diff(prob.row_ptr))
        for e in range(prob.n_edges):
            key = (int(rows[e]), int(prob.col[e]))
            assert np.abs(prob.blk[e].reshape(3, 3) - M[key]).max() < 1e-13 * max(1.0, np.abs(M[key]).max())
            assert prob.w[e] == float(w[key])                      # bit for bit
        assert np.array_equal(prob.deg_c, dc.astype(np.float64)) and np.array_equal(prob.deg_t, dtt.astype(np.float64))


def test_weighted_rotations_rounding_is_spelled_out():
    rng = np.random.default_rng(0)
    kr, R, B = rng.uniform(0.1, 5, 50), rng.standard_normal((50, 3, 3)), rng.standard_normal((50, 3, 3))
    out = frontend.weighted_rotations(kr, R, B)
    A = kr[:, None, None] * R
    for e in (0, 17, 49):
        for i in range(3):
            for j in range(3):
                assert out[e, i, j] == (A[e, i, 0] * B[e, 0, j] + A[e, i, 1] * B[e, 1, j]) + A[e, i, 2] * B[e, 2, j]


class _FakeBackend:
    def __init__(self, abort_on):
        self.abort_on, self.calls, self.failed, self.flag = set(abort_on), 0, [], False

    def synchronize(self):
        pass

    def barrier_aborted(self):
        return self.flag

    def cooperative_failed(self, which):
        self.failed.append(which); self.flag = False


class _Comm:
    def __init__(self, world):
        self.world, self.rank = world, 0


def test_cooperative_fallback_reruns_once_on_the_launch_sequence_path():
    K = _FakeBackend(abort_on=[1])

    def fn():
        K.calls += 1
        if K.calls in K.abort_on:
            K.flag = True
            raise ArithmeticError("garbage from the aborted launch")
        return "result %d" % K.calls
    assert with_cooperative_fallback(K, _Comm(1), fn) == "result 2" and K.failed == ["grid barrier timeout"]
    # no abort: untouched; an error without an abort is the caller's error
    assert with_cooperative_fallback(K, _Comm(1), fn) == "result 3" and len(K.failed) == 1
    with pytest.raises(ZeroDivisionError):
        with_cooperative_fallback(K, _Comm(1), lambda: 1 // 0)
    # sharded runs cannot re-run one rank alone: loud error
    K2 = _FakeBackend(abort_on=[1])

    def fn2():
        K2.calls += 1
        K2.flag = True
        return 0
    with pytest.raises(RuntimeError, match="grid barrier"):
        with_cooperative_fallback(K2, _Comm(2), fn2)
    # stand-in backends without the hooks pass straight through
    assert with_cooperative_fallback(object(), None, lambda: 7) == 7


def test_self_movement_recorder_on_a_small_singular_system():
    """SelfMovement around the oracle's cg: the unperturbed answer is what the oracle returns, the trials move it by a few
    ulps on a well-conditioned graph, and more_trials(rel) scales with rel."""
    import types
    import scipy.sparse as sp
    from scipy.sparse.linalg import cg
    n = 30
    rng = np.random.default_rng(1)
    i, j = rng.integers(0, n, 120), rng.integers(0, n, 120)
    keep = i != j
    W = sp.coo_matrix((np.ones(keep.sum()), (i[keep], j[keep])), shape=(n, n)); W = W + W.T
    L = sp.kron(sp.diags(np.asarray(W.sum(1)).ravel()) - W, sp.eye(3)).tocsr()
    b = L @ rng.standard_normal(3 * n)
    orc = types.SimpleNamespace(cg=cg)
    with SelfMovement(orc, n_trials=4) as sm:
        x, code = orc.cg(L, b)
    assert orc.cg is cg and code == 0 and np.allclose(x, cg(L, b)[0])
    assert sm.self_move.shape == (4,) and sm.self_move.max() < 1e-9 and len(sm.iters) == 5
    assert sm.bound() == 1e-6
    big = sm.more_trials(1e-3, n_trials=3)
    assert big.shape == (3,) and big.max() > 100 * max(sm.self_move.max(), 1e-16)


def _object_scene(n_time=60, n_marker=7, mpv=3, seed=5):
    scene = synth.make_scene(n_cam=1, n_time=n_time, n_marker=n_marker, seed=seed)
    flat = synth.make_object_edges(scene, mpv=mpv, sigma_r=1e-3, sigma_t=1e-3, seed=seed + 1)
    return synth.edges_to_dict(flat, SE3)


def _same_problem(p1, p2):
    for f in ("row_ptr", "col", "blk", "a", "w", "u", "v", "deg_c", "deg_t", "cam_names", "time_names", "tnodes", "tnode_of_cam",
              "tnode_of_time", "src_cam", "src_time", "src_t", "src_qtau", "src_kt"):
        assert np.array_equal(getattr(p1, f), getattr(p2, f)), f
    assert p1.root == p2.root and p1.n_src == p2.n_src


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_object_front_end_one_pass_equals_the_reference_shaped_path(dtype):
    """flatten_object (one pass, batched float32 inversion - what object_bipartite_se3sync runs) against
    invert_object_edges + flatten (a pose.inv() call and a dict copy per edge, as bipgo.py:523-531): the same Problem to the
    bit - with trivial callables, with callables that look at the INVERTED pose / test membership / iterate the value dict,
    with a filter that drops edges (the root is still the smallest id over ALL source edges), and with poses whose R / t
    are float32 views (SE3(pose=...)) instead of float64 arrays."""
    src = _object_scene()
    unit, keep = (lambda e: 1.0), (lambda e: True)
    w_pose = lambda e: 0.5 + float(np.abs(e["pose"].t()).sum())
    f_pose = lambda e: ("pose" in e and e.get("pose") is not None and len(e) == 4 and sorted(e.keys()) == ["corners", "im_filename", "pose", "reprojected_err"]
                        and e["pose"].R()[0, 0] > -0.95)
    f_drop = lambda e: e["reprojected_err"] < np.median([v["reprojected_err"] for v in src.values()])
    src32 = {k: dict(v, pose=SE3(pose=np.block([[v["pose"].R(), v["pose"].t()[:, None]], [np.zeros((1, 3)), np.ones((1, 1))]])))
             for k, v in src.items()}
    for edges, nr, nt, ff in ((src, unit, unit, keep), (src, w_pose, w_pose, f_pose), (src, unit, w_pose, f_drop), (src32, w_pose, unit, keep)):
        root, re_keyed = frontend.invert_object_edges(edges)
        p1 = frontend.flatten(re_keyed, {root: SE3(pose=np.eye(4))}, nr, nt, ff, dtype)
        root2, p2 = frontend.flatten_object(edges, nr, nt, ff, dtype)
        assert root == root2
        _same_problem(p1, p2)


def _dense_blocks(C, T, cam, time, blocks):
    D = np.zeros((C, T, 3, 3))
    D[cam, time] = blocks
    return D


@pytest.mark.parametrize("weight", ["python_float", "numpy_float64", "numpy_float32", "python_int"])
def test_object_mode_blocks_carry_numpys_float32_product(weight):
    """Object mode weights float32 rotations (SE3.inv(), geometry.py:239-243): with a Python scalar numpy forms `k_r * R`
    (bipgo.py:213) in float32 - weight AND product rounded - with a numpy float64 scalar in float64.  The merged blocks of the
    product's front-end equal the pinned oracle's (which performs the reference's own expression edge by edge) TO THE BIT in every
    case; before round 5 the product formed them in float64 always: 4e-8 relative, the 1e-9 ... 6e-8 rad of object-mode
    float64 rotation offset in the random campaign (tools/object_offset_probe.py)."""
    from oracle import bipgo_oracle as orc
    src = _object_scene(n_time=40, n_marker=9, mpv=4, seed=11)
    area = gc.CALLABLES["w_area_mild"]
    nr = {"python_float": area, "numpy_float64": (lambda e: np.float64(area(e))), "numpy_float32": (lambda e: np.float32(area(e))),
          "python_int": (lambda e: 1 + int(area(e) * 7) % 3)}[weight]
    nt, ff = gc.CALLABLES["w_area_mild_t"], gc.CALLABLES["f_all"]
    root, prob = frontend.flatten_object(src, nr, nt, ff, np.float64)
    # the oracle's path: invert edge by edge, then the reference's per-edge expression (oracle/bipgo_oracle.py flatten_edges)
    edges = {}
    for k, v in src.items():
        ts, mid = k[1].split("_")
        edges[(mid, ts + "_" + root)] = dict(v, pose=v["pose"].inv())
    mg = orc.merge_edges(orc.flatten_edges(edges, {root: SE3(pose=np.eye(4))}, nr, nt, ff))
    rows = np.repeat(np.arange(prob.n_time), np.diff(prob.row_ptr))
    mine = _dense_blocks(prob.n_cam, prob.n_time, prob.col, rows, prob.blk.reshape(-1, 3, 3))
    ref = _dense_blocks(mg["C"], mg["T"], mg["cam"], mg["time"], mg["blocks"])
    assert np.array_equal(mine, ref), np.abs(mine - ref).max()
    # ... and it is a float32 product exactly when numpy says so
    is32 = (nr(next(iter(src.values()))) * np.zeros(1, np.float32)).dtype == np.float32
    assert is32 == (weight != "numpy_float64")
    blk = prob.blk[prob.a > 0]
    assert bool(np.all(blk == blk.astype(np.float32))) == is32


def test_float32_poses_in_camera_mode_take_the_same_rule():
    """Camera mode with poses built from a 4x4 matrix (float32 R): the weighted rotation is numpy's float32 product there too
    (then multiplied by float64 constraint rotations in float64, as numpy promotes); float64 poses are untouched."""
    from oracle import bipgo_oracle as orc
    scene = synth.make_scene(n_cam=5, n_time=30, n_marker=4, seed=3)
    flat = synth.make_camera_edges(scene, cpt=3, mpv=2, sigma_r=1e-3, sigma_t=1e-3, seed=4)
    src = synth.edges_to_dict(flat, SE3)
    cons = synth.constraints_from_scene(scene, SE3)
    src32 = {k: dict(v, pose=SE3(pose=np.block([[v["pose"].R(), v["pose"].t()[:, None]], [np.zeros((1, 3)), np.ones((1, 1))]])))
             for k, v in src.items()}
    nr, nt, ff = gc.CALLABLES["w_area_mild"], gc.CALLABLES["w_area_mild_t"], gc.CALLABLES["f_all"]
    for edges, tol in ((src32, 1e-13), (src, 1e-13)):
        prob = frontend.flatten(edges, cons, nr, nt, ff, np.float64)
        mg = orc.merge_edges(orc.flatten_edges(edges, cons, nr, nt, ff))
        rows = np.repeat(np.arange(prob.n_time), np.diff(prob.row_ptr))
        mine = _dense_blocks(prob.n_cam, prob.n_time, prob.col, rows, prob.blk.reshape(-1, 3, 3))
        ref = _dense_blocks(mg["C"], mg["T"], mg["cam"], mg["time"], mg["blocks"])
        # (the constraint products run in another order than BLAS's: 1e-16 relative; a float64 product of float32 poses would
        #  show 6e-8)
        assert np.abs(mine - ref).max() <= tol * np.abs(ref).max(), np.abs(mine - ref).max()


def test_object_front_end_mixed_pose_dtypes_take_the_per_pose_path():
    src = _object_scene(n_time=12)
    keys = list(src)
    v = src[keys[3]]
    src[keys[3]] = dict(v, pose=SE3(pose=np.block([[v["pose"].R(), v["pose"].t()[:, None]], [np.zeros((1, 3)), np.ones((1, 1))]])))
    assert frontend.inverted_poses([e["pose"].R() for e in src.values()], [e["pose"].t() for e in src.values()]) is None
    unit, keep = (lambda e: 1.0), (lambda e: True)
    root, re_keyed = frontend.invert_object_edges(src)
    p1 = frontend.flatten(re_keyed, {root: SE3(pose=np.eye(4))}, unit, unit, keep, np.float64)
    _, p2 = frontend.flatten_object(src, unit, unit, keep, np.float64)
    _same_problem(p1, p2)


def test_string_ids_sort_like_the_reference_on_the_fast_and_the_general_path():
    """sorted_codes: packed-key path (ASCII ids of <= 8 characters, straight from a Python list) and np.unique fallback (longer /
    non-ASCII ids) both reproduce np.unique's string order and inverse."""
    rng = np.random.default_rng(3)
    short = [str(int(x)) for x in rng.integers(0, 3000, 500)] + ["", "a", "Z9", "12345678"]
    long_ = short + ["123456789", "camera-with-a-long-name"]
    uni = short + ["caméra"]
    for ids in (short, long_, uni, np.array(short), np.array(long_)):
        names, inv = frontend.sorted_codes(ids)
        ref_names, ref_inv = np.unique(np.asarray(ids).astype(str), return_inverse=True)
        assert np.array_equal(names, ref_names) and np.array_equal(inv, ref_inv)


def test_c_passes_of_the_front_end_equal_the_python_path(monkeypatch):
    """csrc/vican_fastpath.c (one C pass over the value dicts per column) against the list-comprehension path it replaces: the same
    Problem to the bit for float64 poses, float32 poses (views of a 4x4: strided), a mixture, and foreign pose classes (methods
    instead of attributes); values it does not recognise (dict subclasses, poses whose R() is a list) fall back silently."""
    from vican_amd import _lib
    _lib.build_fastpath()
    monkeypatch.setattr(_lib, "_fastpath", False)                   # (re-probe under the current environment)
    fp = _lib.fastpath()
    assert fp is not None
    scene = synth.make_scene(n_cam=5, n_time=40, n_marker=4, seed=3)
    flat = synth.make_camera_edges(scene, cpt=3, mpv=2, sigma_r=1e-3, sigma_t=1e-3, seed=4)
    src = synth.edges_to_dict(flat, SE3)
    cons = synth.constraints_from_scene(scene, SE3)
    p4 = lambda v: np.block([[v["pose"].R(), v["pose"].t()[:, None]], [np.zeros((1, 3)), np.ones((1, 1))]])

    class Foreign:                                                   # a pose class of the caller's own: R() / t() only
        def __init__(self, R, t):
            self.rot, self.tr = R, t

        def R(self):
            return self.rot

        def t(self):
            return self.tr

        def inv(self):
            return SE3(R=self.rot, t=self.tr).inv()

    variants = {
        "f64": src,
        "f32": {k: dict(v, pose=SE3(pose=p4(v))) for k, v in src.items()},
        "mixed": {k: dict(v, pose=SE3(pose=p4(v)) if i % 3 == 0 else v["pose"]) for i, (k, v) in enumerate(src.items())},
        "foreign": {k: dict(v, pose=Foreign(v["pose"].R(), v["pose"].t())) for k, v in src.items()},
    }
    area = vectorized_area = frontend.vectorized(lambda cols: 0.5 + np.abs(cols["corners"]).sum((1, 2)) / 4000.0 + 0.0 * cols["reprojected_err"])(
        lambda e: 0.5 + float(np.abs(e["corners"]).sum()) / 4000.0)
    nt, ff = gc.CALLABLES["w_area_mild_t"], gc.CALLABLES["f_err"]
    for name, edges in variants.items():
        cols = frontend.EdgeColumns(list(edges.values()))
        assert cols._gather_poses_c(), name                         # the C pass recognises all four
        p_c = frontend.flatten(edges, cons, area, nt, ff, np.float64)
        monkeypatch.setenv("VICAN_FASTPATH", "0"); monkeypatch.setattr(_lib, "_fastpath", False)
        assert _lib.fastpath() is None
        p_py = frontend.flatten(edges, cons, area, nt, ff, np.float64)
        monkeypatch.delenv("VICAN_FASTPATH"); monkeypatch.setattr(_lib, "_fastpath", False)
        _same_problem(p_c, p_py)
    # not recognised: a dict subclass, a pose whose R() is a nested list - the Python path serves, same result as arrays would give
    class D(dict):
        pass
    odd = {k: D(v) for k, v in src.items()}
    assert not frontend.EdgeColumns(list(odd.values()))._gather_poses_c()
    _same_problem(frontend.flatten(odd, cons, area, nt, ff, np.float64), frontend.flatten(src, cons, area, nt, ff, np.float64))
    lists = {k: dict(v, pose=Foreign(v["pose"].R().tolist(), v["pose"].t().tolist())) for k, v in src.items()}
    assert not frontend.EdgeColumns(list(lists.values()))._gather_poses_c()
    _same_problem(frontend.flatten(lists, cons, area, nt, ff, np.float64), frontend.flatten(src, cons, area, nt, ff, np.float64))
    # columns of scalars / small arrays
    cols = frontend.EdgeColumns(list(src.values()))
    assert np.array_equal(cols["corners"], np.array([v["corners"] for v in src.values()], dtype=np.float64))
    assert np.array_equal(cols["reprojected_err"], np.array([v["reprojected_err"] for v in src.values()], dtype=np.float64))


def test_column_form_weights_round_like_the_scalar_form_on_float32_rotations():
    """numpy forms the reference's `k_r * R` (bipgo.py:213) in float32 when R is a float32 array (every SE3 built from a 4x4
    matrix) and the weight a Python float: the column form of such a callable - one float64 array for all edges - must give
    the same merged blocks to the bit (round-5 advisor finding: it formed float64 products, 7e-8 off), and a callable
    that declares NumPy-scalar results (python_scalars=False) the float64 ones."""
    g = load_golden("g2_small")
    case, src, cons, (nr, nt, ff) = rebuild_inputs("g2_small", g)

    def p4(v):
        P = np.eye(4)
        P[:3, :3], P[:3, 3] = v["pose"].R(), v["pose"].t()
        return P
    src32 = {k: dict(v, pose=SE3(pose=p4(v))) for k, v in src.items()}
    assert next(iter(src32.values()))["pose"].R().dtype == np.float32
    # (a weight both forms evaluate with IEEE operations only: math.exp and np.exp may differ in the last place)
    scalar = lambda e: 1.0 / (1.0 + float(e["reprojected_err"]))
    column = frontend.vectorized(lambda cols: 1.0 / (1.0 + cols["reprojected_err"]))(lambda e: 1.0 / (1.0 + float(e["reprojected_err"])))
    p_s = frontend.flatten(src32, cons, scalar, nt, ff, np.float64)
    p_c = frontend.flatten(src32, cons, column, nt, ff, np.float64)
    assert np.array_equal(np.asarray(p_s.blk), np.asarray(p_c.blk)) and np.array_equal(np.asarray(p_s.a), np.asarray(p_c.a))
    # NumPy-scalar weights (np.float64 is not a weak scalar: float64 products) - scalar and column form agree there too
    scalar_np = lambda e: np.float64(1.0) / (1.0 + np.float64(e["reprojected_err"]))
    column_np = frontend.vectorized(lambda cols: 1.0 / (1.0 + cols["reprojected_err"]), python_scalars=False)(scalar_np)
    q_s = frontend.flatten(src32, cons, scalar_np, nt, ff, np.float64)
    q_c = frontend.flatten(src32, cons, column_np, nt, ff, np.float64)
    assert np.array_equal(np.asarray(q_s.blk), np.asarray(q_c.blk))
    assert not np.array_equal(np.asarray(q_s.blk), np.asarray(p_s.blk))          # (the two roundings do differ on this data)
