"""Device-side merge of source edges (vican_merge.hip, reference bipgo.py:203-221 and 445-469) against the host front-end
(frontend.merge_host) - BIT-identical: every golden case in both dtypes, random scenes with many markers per view (long
segments), and the large_shop-sized edge set; then the drop-in call through both merges."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import golden_cases as gc                                    # noqa: E402
from util import rebuild_inputs                              # noqa: E402
from vican_amd import frontend, synth                        # noqa: E402
from vican_amd.geometry import SE3                           # noqa: E402

FIELDS = ("row_ptr", "col", "blk", "a", "w", "u", "v", "deg_c", "deg_t")


def edge_arrays(src, nr, nt, ff):
    cams, times, marks, poses, kr, kt = [], [], [], [], [], []
    for key, val in src.items():
        if not ff(val):
            continue
        ts, mid = key[1].split("_")
        cams.append(key[0]); times.append(ts); marks.append(mid); poses.append(val["pose"])
        kr.append(nr(val)); kt.append(nt(val))
    R = np.stack([np.asarray(p.R(), dtype=np.float64) for p in poses])
    t = np.stack([np.asarray(p.t(), dtype=np.float64).reshape(3) for p in poses])
    return cams, times, marks, R, t, np.asarray(kr, dtype=np.float64), np.asarray(kt, dtype=np.float64)


def assert_same_bits(ix, R, t, kr, kt, dtype, kr_f32=None):
    from vican_amd.device import merge_edges
    h = frontend.merge_host(ix, R, t, kr, kt, dtype, kr_f32=kr_f32)
    d = merge_edges(ix, R, t, kr, kt, dtype, kr_f32=kr_f32)
    assert d.on_device and d.n_edges == h.n_edges and d.n_cam == h.n_cam and d.n_time == h.n_time
    for f in FIELDS:
        a, b = np.asarray(getattr(h, f)), getattr(d, f).cpu().numpy()
        assert a.shape == b.shape, f
        if a.dtype.kind == "f":
            assert np.array_equal(a.view(np.int64), b.astype(np.float64).view(np.int64)), "%s differs (max %.3e)" % (f, np.abs(a - b).max())
        else:
            assert np.array_equal(a, b), f
    assert np.array_equal(d.row_ptr_host, h.row_ptr) and np.array_equal(d.col_host, h.col)
    return h, d


@pytest.mark.parametrize("name", sorted(gc.CASES))
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_device_merge_equals_host_merge_on_goldens(name, dt):
    case, src, cons, (nr, nt, ff) = rebuild_inputs(name)
    if case["mode"] == "object":
        root, src = frontend.invert_object_edges(src)
        cons = {root: SE3(pose=np.eye(4))}
    cams, times, marks, R, t, kr, kt = edge_arrays(src, nr, nt, ff)
    assert_same_bits(frontend.index_edges(cams, times, marks, cons), R, t, kr, kt, dt)


@pytest.mark.parametrize("seed", range(4))
def test_device_merge_long_segments_and_heavy_weights(seed):
    """Up to 24 markers per (camera, timestep) pair (segments long enough for a pairwise summation to differ from the
    sequential one), log-normal weights over eight decades, both dtypes."""
    rng = np.random.default_rng(seed)
    scene = synth.make_scene(n_cam=7, n_time=60, n_marker=24, seed=seed)
    flat = synth.make_camera_edges(scene, cpt=3, mpv=int(rng.integers(9, 25)), sigma_r=1e-2, sigma_t=1e-2, seed=seed + 1)
    src, cons = synth.edges_to_dict(flat, SE3), synth.constraints_from_scene(scene, SE3)
    cams, times, marks, R, t, _, _ = edge_arrays(src, lambda e: 1.0, lambda e: 1.0, lambda e: True)
    kr, kt = np.exp(rng.normal(0, 3.0, len(cams))), np.exp(rng.normal(0, 3.0, len(cams)))
    ix = frontend.index_edges(cams, times, marks, cons)
    for dt in (np.float32, np.float64):
        assert_same_bits(ix, R, t, kr, kt, dt)


def test_device_merge_float32_products_where_flagged():
    """kr_f32 (frontend.f32_product_mask): where set, weight and product k_r * R are rounded to float32 as numpy does for a
    float32 rotation and a Python scalar - host and device agree to the bit, flagged and unflagged edges mixed."""
    rng = np.random.default_rng(8)
    scene = synth.make_scene(n_cam=6, n_time=200, n_marker=5, seed=8)
    flat = synth.make_camera_edges(scene, cpt=3, mpv=3, sigma_r=1e-3, sigma_t=1e-3, seed=9)
    src, cons = synth.edges_to_dict(flat, SE3), synth.constraints_from_scene(scene, SE3)
    cams, times, marks, R, t, _, _ = edge_arrays(src, lambda e: 1.0, lambda e: 1.0, lambda e: True)
    ix = frontend.index_edges(cams, times, marks, cons)
    n = len(cams)
    R = R.astype(np.float32).astype(np.float64)
    kr, kt = rng.uniform(0.3, 3.0, n), rng.uniform(0.5, 2.0, n)
    for mask in (np.ones(n, dtype=bool), rng.random(n) < 0.5):
        for dt in (np.float32, np.float64):
            h, _ = assert_same_bits(ix, R, t, kr, kt, dt, kr_f32=mask)
        plain = frontend.merge_host(ix, R, t, kr, kt, np.float64)
        assert not np.array_equal(h.blk, plain.blk)                 # (the flags do something)


def test_device_merge_at_large_shop_size_and_its_cost():
    scene, flat = gc.build_flat(gc.LARGE_SHOP)
    cons = synth.constraints_from_scene(scene, SE3)
    cams = flat["cam_key"].astype(str)
    tm = np.char.partition(flat["marker_key"].astype(str), "_")
    rng = np.random.default_rng(0)
    kr, kt = rng.uniform(0.5, 2.0, len(cams)), rng.uniform(0.5, 2.0, len(cams))
    t0 = time.perf_counter()
    ix = frontend.index_edges(cams, tm[:, 0], tm[:, 2], cons)
    t_index = time.perf_counter() - t0
    from vican_amd.device import merge_edges
    h, d = assert_same_bits(ix, flat["R"], flat["t"], kr, kt, np.float32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        merge_edges(ix, flat["R"], flat["t"], kr, kt, np.float32)
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    frontend.merge_host(ix, flat["R"], flat["t"], kr, kt, np.float32)
    t_host = time.perf_counter() - t0
    print("large_shop-sized edge set (%d source edges -> %d merged): string ids -> indices %.1f ms (host), merge on the device %.2f ms "
          "(upload of the per-edge arrays included), merge on the host %.1f ms" % (len(cams), h.n_edges, 1e3 * t_index, 1e3 * t_dev, 1e3 * t_host))
    assert t_dev < 0.02


def test_dropin_is_the_same_through_both_merges(monkeypatch):
    """The drop-in API merges on the device by default; VICAN_HOST_MERGE=1 keeps the NumPy merge: same bits in, same poses out."""
    from vican_amd.bipgo import bipartite_se3sync
    case, src, cons, (nr, nt, ff) = rebuild_inputs("g3_medium")
    a = bipartite_se3sync(src, cons, nr, nt, ff, gc.MAXITER, "conjugate_gradient", np.float32)
    monkeypatch.setenv("VICAN_HOST_MERGE", "1")
    b = bipartite_se3sync(src, cons, nr, nt, ff, gc.MAXITER, "conjugate_gradient", np.float32)
    assert list(a) == list(b)
    for k in a:
        assert np.array_equal(a[k].R(), b[k].R()) and np.array_equal(a[k].t(), b[k].t())
