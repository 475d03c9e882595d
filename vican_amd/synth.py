"""Synthetic bipartite camera x timestep pose graphs.

The real datasets (small_room / large_shop / cube_calib) are not in the
reference tree, so tests, golden fixtures and the benchmark use structurally
faithful stand-ins: static cameras, one moving object carrying M markers, each
timestep observed by ``cpt`` cameras that each see ``mpv`` markers, PnP-style
noisy relative poses.  Measurement model (what reference cam.py:173-185 emits):

    edge (c, "t_m").pose = T_c^-1 . T_t . S_m     (marker m at time t, in camera c)

with T_c = world<-camera, T_t = world<-object, S_m = object<-marker.

Two levels:
  * numpy generators producing *source edges* (flat arrays and the reference's
    dict format) for tests / goldens / CPU-scale runs;
  * a torch generator producing *merged* (camera,timestep) blocks directly on
    the GPU for the HBM-bound stress configuration (1e7..1e8 edges), where
    materialising a Python dict is out of the question.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np

# --------------------------------------------------------------------------
# small SO(3) utilities (numpy, batched)
# --------------------------------------------------------------------------

def random_rotations(rng: np.random.Generator, n: int) -> np.ndarray:
    """Haar-distributed rotations from normalised Gaussian quaternions."""
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    r = np.empty((n, 3, 3))
    r[:, 0, 0] = 1 - 2 * (y * y + z * z)
    r[:, 0, 1] = 2 * (x * y - z * w)
    r[:, 0, 2] = 2 * (x * z + y * w)
    r[:, 1, 0] = 2 * (x * y + z * w)
    r[:, 1, 1] = 1 - 2 * (x * x + z * z)
    r[:, 1, 2] = 2 * (y * z - x * w)
    r[:, 2, 0] = 2 * (x * z - y * w)
    r[:, 2, 1] = 2 * (y * z + x * w)
    r[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return r


def exp_so3(w: np.ndarray) -> np.ndarray:
    """Rodrigues formula, batched over leading dim: (n,3) -> (n,3,3)."""
    th = np.linalg.norm(w, axis=1)
    small = th < 1e-12
    ths = np.where(small, 1.0, th)
    k = w / ths[:, None]
    kx = np.zeros((len(w), 3, 3))
    kx[:, 0, 1], kx[:, 0, 2] = -k[:, 2], k[:, 1]
    kx[:, 1, 0], kx[:, 1, 2] = k[:, 2], -k[:, 0]
    kx[:, 2, 0], kx[:, 2, 1] = -k[:, 1], k[:, 0]
    s = np.where(small, 0.0, np.sin(th))[:, None, None]
    c1 = np.where(small, 0.0, 1.0 - np.cos(th))[:, None, None]
    return np.eye(3)[None] + s * kx + c1 * (kx @ kx)


# --------------------------------------------------------------------------
# scene + source edges (numpy)
# --------------------------------------------------------------------------

def make_scene(n_cam: int, n_time: int, n_marker: int, seed: int = 0,
               cam_spread: float = 5.0, marker_spread: float = 0.3,
               cam_ids=None, time_ids=None, marker_ids=None) -> Dict[str, np.ndarray]:
    """Ground truth: camera poses, object trajectory, marker poses on the object.
    Marker 0 (or the first id) is the identity so the object frame is its frame."""
    rng = np.random.default_rng(seed)
    sc = {
        "R_cam": random_rotations(rng, n_cam),
        "p_cam": cam_spread * rng.standard_normal((n_cam, 3)),
        "R_obj": random_rotations(rng, n_time),
        "p_obj": cam_spread * rng.standard_normal((n_time, 3)),
        "R_mk": random_rotations(rng, n_marker),
        "q_mk": marker_spread * rng.standard_normal((n_marker, 3)),
    }
    sc["R_mk"][0] = np.eye(3)
    sc["q_mk"][0] = 0.0
    sc["cam_ids"] = np.array([str(i) for i in range(n_cam)] if cam_ids is None else list(map(str, cam_ids)))
    sc["time_ids"] = np.array([str(i) for i in range(n_time)] if time_ids is None else list(map(str, time_ids)))
    sc["marker_ids"] = np.array([str(i) for i in range(n_marker)] if marker_ids is None else list(map(str, marker_ids)))
    return sc


def _visibility(rng, n_cam, n_time, cpt):
    """(T, cpt) camera indices per timestep: cameras t%C and (t+1)%C (keeps the
    camera graph connected), the rest random without repetition."""
    cpt = min(cpt, n_cam)
    t = np.arange(n_time)
    vis = np.empty((n_time, cpt), dtype=np.int64)
    vis[:, 0] = t % n_cam
    if cpt > 1:
        vis[:, 1] = (t + 1) % n_cam
    if cpt > 2:
        # random others: draw a random permutation offset, skip the two fixed ones
        scores = rng.random((n_time, n_cam))
        scores[t, vis[:, 0]] = 2.0
        scores[t, vis[:, 1]] = 2.0
        vis[:, 2:] = np.argsort(scores, axis=1)[:, : cpt - 2]
    return vis


def make_camera_edges(scene, cpt: int = 3, mpv: int = 2, sigma_r: float = 1e-3,
                      sigma_t: float = 1e-3, seed: int = 1) -> Dict[str, np.ndarray]:
    """Source edges of the camera-calibration problem (static cameras, moving
    object).  Returns flat arrays; ``edges_to_dict`` makes the reference format."""
    rng = np.random.default_rng(seed)
    C, T, M = len(scene["R_cam"]), len(scene["R_obj"]), len(scene["R_mk"])
    cpt, mpv = min(cpt, C), min(mpv, M)
    vis = _visibility(rng, C, T, cpt)                       # (T,cpt)
    mk_scores = rng.random((T, cpt, M))
    mk = np.argsort(mk_scores, axis=2)[:, :, :mpv]          # (T,cpt,mpv)
    ti = np.broadcast_to(np.arange(T)[:, None, None], mk.shape).reshape(-1)
    ci = np.broadcast_to(vis[:, :, None], mk.shape).reshape(-1)
    mi = mk.reshape(-1)
    return _measure(scene, ci, ti, mi, sigma_r, sigma_t, rng, mode="camera")


def make_object_edges(scene, mpv: int = 4, sigma_r: float = 1e-4,
                      sigma_t: float = 1e-4, seed: int = 1) -> Dict[str, np.ndarray]:
    """Source edges of the object-calibration problem (one moving camera looking
    at the marker cube): keys are (t, "t_m") as in reference bipgo.py:509-515.
    The scene's single camera path is taken from R_obj/p_obj inverted, i.e. the
    object is static and 'time' indexes camera poses."""
    rng = np.random.default_rng(seed)
    T, M = len(scene["R_obj"]), len(scene["R_mk"])
    mpv = min(mpv, M)
    # consecutive frames share markers so the marker graph is connected
    base = (np.arange(T) * 1) % M
    offs = np.arange(mpv)
    mk = (base[:, None] + offs[None, :]) % M                # (T,mpv)
    ti = np.repeat(np.arange(T), mpv)
    mi = mk.reshape(-1)
    ci = np.zeros_like(ti)
    return _measure(scene, ci, ti, mi, sigma_r, sigma_t, rng, mode="object")


def _measure(scene, ci, ti, mi, sigma_r, sigma_t, rng, mode):
    E = len(ci)
    if mode == "camera":
        Rc, pc = scene["R_cam"][ci], scene["p_cam"][ci]
        cam_key = scene["cam_ids"][ci]
    else:  # one moving camera; its pose at frame t is the identity, object moves
        Rc = np.broadcast_to(np.eye(3), (E, 3, 3))
        pc = np.zeros((E, 3))
        cam_key = scene["time_ids"][ti]
    Rt, pt = scene["R_obj"][ti], scene["p_obj"][ti]
    Rm, qm = scene["R_mk"][mi], scene["q_mk"][mi]
    RcT = np.swapaxes(Rc, 1, 2)
    R = RcT @ Rt @ Rm
    t = np.einsum("eij,ej->ei", RcT, pt + np.einsum("eij,ej->ei", Rt, qm) - pc)
    R = R @ exp_so3(sigma_r * rng.standard_normal((E, 3)))
    t = t + sigma_t * rng.standard_normal((E, 3))
    # PnP-like side data: apparent marker square (pixels) shrinks with distance
    dist = np.maximum(np.linalg.norm(t, axis=1), 0.2)
    side = 60.0 / dist * (1.0 + 0.05 * rng.standard_normal(E))
    cx, cy = 300 + 50 * rng.standard_normal(E), 200 + 50 * rng.standard_normal(E)
    sq = np.array([[-0.5, -0.5], [0.5, -0.5], [0.5, 0.5], [-0.5, 0.5]])
    corners = sq[None] * side[:, None, None] + np.stack([cx, cy], 1)[:, None, :]
    err = np.abs(0.01 * rng.standard_normal(E))
    mkey = np.char.add(np.char.add(scene["time_ids"][ti], "_"), scene["marker_ids"][mi])
    return {
        "cam_key": np.asarray(cam_key), "marker_key": mkey,
        "R": R, "t": t, "corners": corners, "reprojected_err": err,
        "cam_idx": ci.astype(np.int64), "time_idx": ti.astype(np.int64),
        "marker_idx": mi.astype(np.int64),
    }


def edges_to_dict(flat: Dict[str, np.ndarray], se3_cls) -> dict:
    """Reference edge-dict format (cam.py:180-185) from flat arrays."""
    out = {}
    ck, mk = flat["cam_key"], flat["marker_key"]
    for e in range(len(ck)):
        out[(str(ck[e]), str(mk[e]))] = {
            "pose": se3_cls(R=flat["R"][e].copy(), t=flat["t"][e].copy()),
            "corners": flat["corners"][e].copy(),
            "reprojected_err": float(flat["reprojected_err"][e]),
            "im_filename": "synthetic",
        }
    return out


def constraints_from_scene(scene, se3_cls) -> dict:
    """``constraints`` argument of bipartite_se3sync: marker id -> object<-marker pose."""
    return {str(m): se3_cls(R=scene["R_mk"][i].copy(), t=scene["q_mk"][i].copy())
            for i, m in enumerate(scene["marker_ids"])}


def shoelace_area(corners: np.ndarray) -> float:
    """Polygon area of a (4,2) corner array - stands in for the notebook's
    ``shapely.Polygon(...).area`` (main.ipynb:75-77,134-136); shapely is absent."""
    x, y = corners[:, 0], corners[:, 1]
    return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))


# --------------------------------------------------------------------------
# merged-block generator on the GPU (stress / bench scale)
# --------------------------------------------------------------------------

def make_merged_graph_torch(n_cam: int, n_time: int, cams_per_t: int, device,
                            dtype, seed: int = 0, sigma_r: float = 1e-2,
                            sigma_t: float = 1e-2, weight_sigma: float = 0.0,
                            t_offset: int = 0, n_time_total: Optional[int] = None):
    """Merged (camera,timestep) edges generated directly as torch tensors on
    ``device`` in timestep-major CSR order (row = timestep, cols ascending).

    Returns a dict with
      row_ptr (T+1,) int32, col (E,) int32,
      blk (E,9) ``dtype``   M_ct = a_ct * R_c^T R_t * noise   (bipgo.py:213-221)
      a (E,) ``dtype``      rotation weight a_ct,
      w (E,) f64            translation weight  sum k_t^2,
      u (E,3) f64           sum k_t^2 * t~_e,  v (E,3) f64 (zero: single root marker),
      R_cam (C,3,3), p_cam (C,3), R_obj (T,3,3), p_obj (T,3) ground truth (f64).

    Every timestep sees cameras t%C and (t+1)%C plus ``cams_per_t-2`` random
    distinct others (same rule as the numpy generator).  ``t_offset`` /
    ``n_time_total`` let each rank of a timestep-sharded run generate only its
    own rows of one global graph.
    """
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    C, T = n_cam, n_time
    k = min(cams_per_t, C)
    f64 = torch.float64

    def rand_rot(n, gen):
        q = torch.randn((n, 4), generator=gen, device=device, dtype=f64)
        q = q / q.norm(dim=1, keepdim=True)
        w, x, y, z = q.unbind(1)
        return torch.stack([
            1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
            2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
            2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).view(n, 3, 3)

    R_cam = rand_rot(C, g)
    p_cam = 5.0 * torch.randn((C, 3), generator=g, device=device, dtype=f64)
    # per-rank rows: reseed per global timestep block so shards are consistent
    g2 = torch.Generator(device=device)
    g2.manual_seed(seed * 1000003 + 17 + t_offset)
    R_obj = rand_rot(T, g2)
    p_obj = 5.0 * torch.randn((T, 3), generator=g2, device=device, dtype=f64)

    tglob = torch.arange(T, device=device, dtype=torch.int64) + t_offset
    cols = torch.empty((T, k), device=device, dtype=torch.int64)
    cols[:, 0] = tglob % C
    if k > 1:
        cols[:, 1] = (tglob + 1) % C
    if k > 2:
        if k >= C:
            cols = torch.arange(C, device=device).expand(T, C).clone()
        else:
            # random distinct others: offsets in [2, C) relative to t%C, distinct
            # via random keys on a (rows x C-2) strip, processed in row batches
            step = max(1, (1 << 24) // max(C, 1))
            for s in range(0, T, step):
                e = min(T, s + step)
                keys = torch.rand((e - s, C - 2), generator=g2, device=device)
                off = torch.argsort(keys, dim=1)[:, : k - 2] + 2
                cols[s:e, 2:] = (tglob[s:e, None] + off) % C
    cols, _ = torch.sort(cols, dim=1)
    col = cols.reshape(-1)
    E = col.numel()
    row = torch.arange(T, device=device, dtype=torch.int64).repeat_interleave(k)
    row_ptr = (torch.arange(T + 1, device=device, dtype=torch.int64) * k).to(torch.int32)

    a = torch.ones(E, device=device, dtype=f64)
    if weight_sigma > 0:
        a = torch.exp(weight_sigma * torch.randn(E, generator=g2, device=device, dtype=f64))
    # noise rotation via Rodrigues on small vectors, in edge batches to bound memory
    blk = torch.empty((E, 9), device=device, dtype=dtype)
    u = torch.empty((E, 3), device=device, dtype=f64)
    bs = 1 << 22
    eye = torch.eye(3, device=device, dtype=f64)
    for s in range(0, E, bs):
        e = min(E, s + bs)
        wv = sigma_r * torch.randn((e - s, 3), generator=g2, device=device, dtype=f64)
        th = wv.norm(dim=1).clamp_min(1e-30)
        kk = wv / th[:, None]
        K = torch.zeros((e - s, 3, 3), device=device, dtype=f64)
        K[:, 0, 1], K[:, 0, 2] = -kk[:, 2], kk[:, 1]
        K[:, 1, 0], K[:, 1, 2] = kk[:, 2], -kk[:, 0]
        K[:, 2, 0], K[:, 2, 1] = -kk[:, 1], kk[:, 0]
        N = eye + torch.sin(th)[:, None, None] * K + (1 - torch.cos(th))[:, None, None] * (K @ K)
        RcT = R_cam[col[s:e]].transpose(1, 2)
        Rt = R_obj[row[s:e]]
        M = (RcT @ Rt @ N) * a[s:e, None, None]
        blk[s:e] = M.reshape(-1, 9).to(dtype)
        tt = torch.einsum("eij,ej->ei", RcT, p_obj[row[s:e]] - p_cam[col[s:e]])
        tt = tt + sigma_t * torch.randn((e - s, 3), generator=g2, device=device, dtype=f64)
        u[s:e] = tt * (a[s:e, None] ** 2)
    # (a benchmark must never time a solve of garbage: round 6 met generated graphs with NaN blocks - memory handed out after a
    #  freed uncached allocation, csrc/vican_comm.hip "Mailboxes are NEVER handed back" - and the solver dutifully "failed to converge")
    if not bool(torch.isfinite(blk).all()) or not bool(torch.isfinite(u).all()):
        raise FloatingPointError("make_merged_graph_torch: non-finite entries in the generated graph")
    return {
        "row_ptr": row_ptr, "col": col.to(torch.int32), "blk": blk,
        "a": a.to(dtype), "w": a * a, "u": u,
        "v": torch.zeros((E, 3), device=device, dtype=f64),
        "R_cam": R_cam, "p_cam": p_cam, "R_obj": R_obj, "p_obj": p_obj,
        "n_cam": C, "n_time": T,
    }


def make_ragged_graph_torch(C, T, lo, hi, dev, seed=0):
    """(row_ptr, col, blk f32, a f32) of a graph whose timestep t is seen by a uniformly random number of cameras in [lo, hi]."""
    import torch
    g = torch.Generator(device=dev); g.manual_seed(seed)
    deg = torch.randint(lo, hi + 1, (T,), generator=g, device=dev)
    idx = torch.rand(T, C, generator=g, device=dev).topk(hi, dim=1).indices                    # hi distinct random cameras per row
    idx = torch.where(torch.arange(hi, device=dev)[None, :] < deg[:, None], idx, torch.full_like(idx, C)).sort(1).values
    col = idx[idx < C].to(torch.int32)
    row_ptr = torch.zeros(T + 1, dtype=torch.int32, device=dev); row_ptr[1:] = deg.cumsum(0)
    E = int(col.numel())
    q = torch.randn(E, 4, generator=g, device=dev, dtype=torch.float64); q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                     2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)
    a = torch.rand(E, generator=g, device=dev, dtype=torch.float64) + 0.5
    return row_ptr, col, (R * a[:, None]).to(torch.float32), a.to(torch.float32)
