"""ORACLE - CPU restatement of the reference's primal-dual bipartite SE(3) solver.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the
checker / the timed CPU baseline.  The product (``vican_amd``) never imports it
and has no CPU fallback.

What it restates (own NumPy/SciPy code, same third-party calls the reference makes):

  ``large_bipartite_so3sync``    reference vican/bipgo.py:145-350
  ``bipartite_se3sync``          reference vican/bipgo.py:353-490
  ``object_bipartite_se3sync``   reference vican/bipgo.py:493-545
  ``bipartite_so3sync``          reference vican/bipgo.py:18-142 (non-eliminated variant; dead code upstream)

Parity status: PINNED.  ``tests/golden/*.npz`` hold outputs of the REAL reference
run in the build container (``tests/golden/make_golden.py``, numpy 2.2.6 /
scipy 1.15.3 - the reference pins numpy 1.19.5 / scipy 1.5.4, requirements.txt:2,7;
the library versions are recorded inside each fixture) and
``tests/test_oracle_golden.py`` checks this module against every one of them.
The reference itself ships no tests or golden vectors.

Layering: ``flatten_edges`` (dict -> arrays; bipgo.py:203-221,420-431,445-455),
``so3sync_arrays`` (rotation stage on arrays; bipgo.py:225-348) and
``translation_arrays`` (bipgo.py:434-480) so kernel tests can compare
stage-by-stage; the dict-level entry points compose them.
"""
from __future__ import annotations

import numpy as np
from scipy.sparse import csr_matrix, diags
from scipy.sparse.linalg import cg, eigs, lsqr

__all__ = ["flatten_edges", "so3sync_arrays", "translation_arrays", "so3sync_general_arrays",
           "bipartite_se3sync", "object_bipartite_se3sync", "bipartite_so3sync", "polar_dual", "Pose"]


class Pose:
    """Minimal pose holder for oracle outputs (R world<-node, t)."""

    def __init__(self, R, t):
        self._R, self._t = R, t

    def R(self):
        return self._R

    def t(self):
        return self._t


# ---------------------------------------------------------------------------
# stage 0: dict -> arrays
# ---------------------------------------------------------------------------

def flatten_edges(src_edges, constraints, noise_model_r, noise_model_t, edge_filter):
    """Apply filter, constraints and weights edge by edge (bipgo.py:203-221 and
    420-455) and return flat arrays over the KEPT source edges, in dict order.

    Node ordering follows the reference: cameras sorted by the string 'c'+id,
    timesteps by 't'+timestamp (np.unique on strings, bipgo.py:225-229), the
    translation unknowns by the mixed list of camera ids and '<t>_0' strings
    (bipgo.py:426-430).
    """
    root = str(min(list(constraints.keys())))                     # bipgo.py:196,411
    r_root = np.asarray(constraints[root].R())
    cam_s, time_s, mk_s, rr, tt, kr, kt, rel_r, rel_t = [], [], [], [], [], [], [], [], []
    cache = {}
    for key, val in src_edges.items():
        if not edge_filter(val):
            continue
        ts, mid = key[1].split("_")
        if mid not in cache:
            cm = constraints[mid]                                  # KeyError as bipgo.py:209
            r_m = np.asarray(cm.R())
            q = np.asarray(r_root).T @ r_m                         # bipgo.py:451
            tau = np.asarray((cm.inv() @ constraints[root]).t())  # bipgo.py:452 (float32 chain)
            cache[mid] = (r_m, q, tau)
        r_m, q, tau = cache[mid]
        w_r = noise_model_r(val)
        pose = val["pose"]
        blk = w_r * np.asarray(pose.R()) @ r_m.T @ r_root          # bipgo.py:213
        cam_s.append(key[0]); time_s.append(ts); mk_s.append(mid)
        rr.append(np.asarray(blk, dtype=np.float64))
        tt.append(np.asarray(pose.t(), dtype=np.float64))
        kr.append(float(w_r)); kt.append(float(noise_model_t(val)))
        rel_r.append(np.asarray(q, dtype=np.float64))
        rel_t.append(np.asarray(tau, dtype=np.float64))
    n = len(cam_s)
    cam_s, time_s = np.array(cam_s, dtype=str), np.array(time_s, dtype=str)
    cam_nodes, cam_idx = np.unique(np.char.add("c", cam_s), return_inverse=True)
    time_nodes, time_idx = np.unique(np.char.add("t", time_s), return_inverse=True)
    tnodes = np.unique(np.concatenate([cam_s, np.char.add(time_s, "_0")]))
    pos = {s: i for i, s in enumerate(tnodes)}
    return dict(
        n=n, root=root,
        cam_names=np.array([c[1:] for c in cam_nodes]), time_names=np.array([t[1:] for t in time_nodes]),
        cam_idx=cam_idx.astype(np.int64), time_idx=time_idx.astype(np.int64),
        marker=np.array(mk_s), wR=np.array(rr).reshape(n, 3, 3), t=np.array(tt).reshape(n, 3),
        k_r=np.array(kr), k_t=np.array(kt),
        rel_R=np.array(rel_r).reshape(n, 3, 3), rel_t=np.array(rel_t).reshape(n, 3),
        tnodes=tnodes,
        tnode_of_cam=np.array([pos[c[1:]] for c in cam_nodes], dtype=np.int64),
        tnode_of_time=np.array([pos[t[1:] + "_0"] for t in time_nodes], dtype=np.int64),
    )


def merge_edges(flat):
    """Sum weighted blocks of source edges sharing (camera,timestep), keeping the
    first-appearance order of merged edges like the reference's dict (bipgo.py:215-221)."""
    C, T = len(flat["cam_names"]), len(flat["time_names"])
    key = flat["cam_idx"] * T + flat["time_idx"]
    _, first, inv = np.unique(key, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")             # merged edges in first-seen order
    rank = np.empty_like(order); rank[order] = np.arange(len(order))
    mid = rank[inv]
    ne = len(order)
    blocks = np.zeros((ne, 3, 3)); a = np.zeros(ne)
    np.add.at(blocks, mid, flat["wR"])
    np.add.at(a, mid, flat["k_r"])
    ci = flat["cam_idx"][first[order]]
    ti = flat["time_idx"][first[order]]
    return dict(C=C, T=T, cam=ci, time=ti, blocks=blocks, a=a, merged_of_src=mid)


# ---------------------------------------------------------------------------
# 3x3 polar / dual blocks
# ---------------------------------------------------------------------------

def polar_dual(mats, loop=False):
    """For each 3x3 block A=U S V^T return (U diag(1,1,det(UV^T)) V^T, U, S)
    (bipgo.py:307-308,324-325; geometry.py:189-190).  ``loop=True`` calls LAPACK
    once per node like the reference (used to time the 'reference-shaped' baseline)."""
    if loop:
        us, ss, vs = [], [], []
        for m in mats:
            u, s, vt = np.linalg.svd(m)
            us.append(u); ss.append(s); vs.append(vt)
        u, s, vt = np.array(us), np.array(ss), np.array(vs)
    else:
        u, s, vt = np.linalg.svd(mats)
    d = np.linalg.det(u @ vt)
    fix = np.ones((len(mats), 3)); fix[:, 2] = d
    return (u * fix[:, None, :]) @ vt, u, s


# ---------------------------------------------------------------------------
# stage 1: rotations (primal-dual iteration)
# ---------------------------------------------------------------------------

def so3sync_arrays(C, T, cam, time, blocks, a, maxiter, dtype=np.float32, loop=False, info=None):
    """Primal-dual SO(3) synchronisation on merged edges (bipgo.py:238-348).

    cam/time: (E,) node indices; blocks: (E,3,3) summed weighted rotations M_ct;
    a: (E,) summed weights.  Returns (Rc, Rt): (C,3,3), (T,3,3) world<-node
    rotations (the transposes taken at bipgo.py:346,348), float64 copies of the
    ``dtype`` computation.
    """
    E = len(cam)
    off_r = np.repeat(np.arange(3), 3)[None, :]
    off_c = np.tile(np.arange(3), 3)[None, :]
    bi = (3 * cam[:, None] + off_r).reshape(-1)
    bj = (3 * time[:, None] + off_c).reshape(-1)
    rct = csr_matrix((blocks.reshape(-1).astype(dtype), (bi, bj)), shape=(3 * C, 3 * T))    # :269
    adj = csr_matrix((a.astype(dtype), (cam, time)), shape=(C, T))                              # :270
    deg_t = np.asarray(adj.sum(axis=0)).squeeze()
    P = rct @ diags(1.0 / np.repeat(deg_t, 3), 0) @ rct.T                                       # :273
    padj = adj @ diags(1.0 / deg_t) @ adj.T
    lam_c = diags(np.repeat(np.asarray(padj.sum(axis=-1)).squeeze(), 3), 0)                     # :274-276
    evals_hist = []
    r_c = r_t = None
    blk_rows = lambda n: (3 * np.arange(n)[:, None] + off_r).reshape(-1)
    blk_cols = lambda n: (3 * np.arange(n)[:, None] + off_c).reshape(-1)
    max_eval = 1.0
    for _ in range(maxiter):
        if max_eval <= 1e-6:                                                                    # :283
            break
        L = lam_c - P
        L = 0.5 * (L.T + L)                                                                     # :285-286
        ev, evec = eigs(L, k=5, sigma=-1e-6)                                                    # :288
        ev, evec = np.real(ev), np.real(evec)
        evals_hist.append(ev.astype(np.float64))
        max_eval = np.abs(ev).max()
        # the reference takes columns 0..2 trusting ARPACK's return order; the
        # restatement picks the 3 algebraically smallest explicitly
        sel = np.argsort(ev)[:3]
        v3 = evec[:, sel]
        x = v3 @ np.linalg.inv(v3[:3, :])                                                       # :295
        x = polar_dual(x.reshape(C, 3, 3), loop)[0].reshape(3 * C, 3).astype(x.dtype)           # :296-297
        y = P @ x                                                                               # :300
        rc, u, s = polar_dual(np.asarray(y).reshape(C, 3, 3), loop)                             # :306-312
        lam_blocks = (u * s[:, None, :]) @ np.swapaxes(u, 1, 2)
        lam_c = csr_matrix((lam_blocks.reshape(-1).astype(dtype), (blk_rows(C), blk_cols(C))),
                           shape=(3 * C, 3 * C))                                                # :314
        r_c = rc.reshape(3 * C, 3).astype(x.dtype)
        z = rct.T @ r_c                                                                         # :318
        rt, u, s = polar_dual(np.asarray(z).reshape(T, 3, 3), loop)                             # :323-329
        lt_blocks = (u * (1.0 / s)[:, None, :]) @ np.swapaxes(u, 1, 2)
        lam_t = csr_matrix((lt_blocks.reshape(-1).astype(dtype), (blk_rows(T), blk_cols(T))),
                           shape=(3 * T, 3 * T))                                                # :331
        r_t = rt
        P = rct @ lam_t @ rct.T                                                                 # :334
    if info is not None:
        info["evals"] = np.array(evals_hist)
    Rc = np.swapaxes(np.asarray(r_c, dtype=np.float64).reshape(C, 3, 3), 1, 2)                  # :346
    Rt = np.swapaxes(np.asarray(r_t, dtype=np.float64).reshape(T, 3, 3), 1, 2)                  # :348
    # the reference stores r_c in the eigenvector dtype and r_t in z's dtype
    return Rc.astype(dtype).astype(np.float64), Rt.astype(dtype).astype(np.float64)


# ---------------------------------------------------------------------------
# stage 2: translations (least squares on the incidence matrix)
# ---------------------------------------------------------------------------

def translation_arrays(n_nodes, node_c, node_t, Rc_e, Rt_e, t_meas, rel_R, rel_t, k_t,
                       lsqr_solver, dtype=np.float32, info=None, loop=False):
    """Per kept source edge e between translation unknowns node_c[e], node_t[e]
    (bipgo.py:434-480):  k_t (p_t - p_c) = k_t (R_c t~ + R_t R_root^T R_m tau_m)."""
    ne = len(node_c)
    if loop:
        # edge by edge in the reference's evaluation order: the loosely converged CG
        # amplifies 1e-15 differences in b to ~1e-5 m on non-unit weights, so the
        # tight pin against the goldens needs bit-identical right-hand sides
        rhs = np.empty((ne, 3))
        for e in range(ne):
            rhs[e] = k_t[e] * (Rc_e[e] @ t_meas[e] + Rt_e[e] @ rel_R[e] @ rel_t[e])             # :454-455
    else:
        rhs = k_t[:, None] * (np.einsum("eij,ej->ei", Rc_e, t_meas) +
                              np.einsum("eij,ej->ei", Rt_e @ rel_R, rel_t))
    b = rhs.reshape(-1)
    rows = (3 * np.arange(ne)[:, None] + np.repeat(np.arange(3), 3)[None]).reshape(-1)
    cc = (3 * node_c[:, None] + np.tile(np.arange(3), 3)[None]).reshape(-1)
    ct = (3 * node_t[:, None] + np.tile(np.arange(3), 3)[None]).reshape(-1)
    eye = np.eye(3, dtype=dtype).reshape(-1)[None]
    dc = (-k_t[:, None] * eye).astype(dtype)                                                    # :465
    dt = (k_t[:, None] * eye).astype(dtype)                                                     # :468
    # same triplet order as the reference (per edge: 9 camera entries, then 9
    # timestep entries, explicit zeros included) - the loosely converged CG is
    # sensitive to the summation order inside J^T J
    il = lambda p, q: np.concatenate([p.reshape(ne, 9), q.reshape(ne, 9)], axis=1).reshape(-1)
    J = csr_matrix((il(dc, dt), (il(rows, rows), il(cc, ct))), shape=(3 * ne, 3 * n_nodes))     # :471
    if lsqr_solver == "conjugate_gradient":
        A, rhs_n = J.T @ J, J.T @ b
        hist = []
        x, code = cg(A, rhs_n, callback=lambda xk: hist.append(1))                              # :477
        assert code == 0                                                                        # :478
        if info is not None:
            info["cg_iters"] = len(hist)
            info["cg_relres"] = float(np.linalg.norm(rhs_n - A @ x) / np.linalg.norm(rhs_n))
    elif lsqr_solver == "direct":
        x = lsqr(J, b)[0]                                                                       # :480
    else:
        raise UnboundLocalError("local variable 't_est' referenced before assignment")          # :487
    return np.asarray(x, dtype=np.float64).reshape(n_nodes, 3)


# ---------------------------------------------------------------------------
# dict-level entry points
# ---------------------------------------------------------------------------

def bipartite_se3sync(src_edges, constraints, noise_model_r, noise_model_t, edge_filter,
                      maxiter, lsqr_solver, dtype=np.float32, loop=False, info=None):
    flat = flatten_edges(src_edges, constraints, noise_model_r, noise_model_t, edge_filter)
    mg = merge_edges(flat)
    Rc, Rt = so3sync_arrays(mg["C"], mg["T"], mg["cam"], mg["time"], mg["blocks"], mg["a"],
                            maxiter, dtype, loop, info)
    x = translation_arrays(len(flat["tnodes"]), flat["tnode_of_cam"][flat["cam_idx"]],
                           flat["tnode_of_time"][flat["time_idx"]],
                           Rc[flat["cam_idx"]], Rt[flat["time_idx"]], flat["t"],
                           flat["rel_R"], flat["rel_t"], flat["k_t"], lsqr_solver, dtype, info, loop)
    out = {}
    rot = {}
    for i, c in enumerate(flat["cam_names"]):
        rot[c] = Rc[i]
    for i, t in enumerate(flat["time_names"]):
        rot[t + "_0"] = Rt[i]
    for i, n in enumerate(flat["tnodes"]):                                                      # :485-487
        out[n] = Pose(rot[n], x[i])
    return out


def object_bipartite_se3sync(src_edges, noise_model_r, noise_model_t, edge_filter,
                             maxiter, lsqr_solver, dtype=np.float32, loop=False, info=None):
    root = str(min(int(k[1].split("_")[1]) for k in src_edges.keys()))                          # :524
    edges = {}
    for k, v in src_edges.items():                                                              # :526-531
        ts, mid = k[1].split("_")
        e = dict(v)
        e["pose"] = v["pose"].inv()
        edges[(mid, ts + "_" + root)] = e
    ident = type(next(iter(src_edges.values()))["pose"])(pose=np.eye(4))
    out = bipartite_se3sync(edges, {root: ident}, noise_model_r, noise_model_t, edge_filter,
                            maxiter, lsqr_solver, dtype, loop, info)
    return {k: v for k, v in out.items() if "_" not in k}                                       # :543


# ---------------------------------------------------------------------------
# the non-eliminated variant (bipgo.py:18-142)
# ---------------------------------------------------------------------------

def so3sync_general_arrays(C, T, cam, time, blocks, a, maxiter, dtype=np.float32, info=None):
    """Primal-dual iteration on all N = C+T nodes (cameras first; bipgo.py:54-133).

    blocks: (E,3,3) = sum k_r R~ R_m R_0^T per merged edge (bipgo.py:45-49), a: (E,) summed weights.
    Returns r (N,3,3): the blocks of the reference's final ``r`` as they are (u @ vt WITHOUT det fix,
    not transposed - bipgo.py:127,139-141)."""
    N = C + T
    i, j = np.asarray(cam), C + np.asarray(time)
    off_r = np.repeat(np.arange(3), 3)[None, :]
    off_c = np.tile(np.arange(3), 3)[None, :]
    bi = np.concatenate([(3 * i[:, None] + off_r), (3 * j[:, None] + off_r)], axis=1).reshape(-1)        # :81-84
    bj = np.concatenate([(3 * j[:, None] + off_c), (3 * i[:, None] + off_c)], axis=1).reshape(-1)
    bd = np.concatenate([blocks.reshape(-1, 9), np.swapaxes(blocks, 1, 2).reshape(-1, 9)], axis=1).reshape(-1)
    pr = csr_matrix((bd.astype(dtype), (bi, bj)), shape=(3 * N, 3 * N))                                   # :91
    adj = csr_matrix((np.repeat(a, 2).astype(dtype), (np.stack([i, j], 1).reshape(-1), np.stack([j, i], 1).reshape(-1))),
                     shape=(N, N))                                                                         # :92
    deg = np.asarray(adj.sum(axis=1)).squeeze()
    lbd = diags(np.repeat(deg, 3).astype(dtype), 0)                                                        # :95-99
    rows = (3 * np.arange(N)[:, None] + off_r).reshape(-1)
    cols = (3 * np.arange(N)[:, None] + off_c).reshape(-1)
    hist = []
    r = None
    for _ in range(maxiter):
        L = lbd - pr
        L = 0.5 * (L.T + L)                                                                                # :102-103
        ev, evec = eigs(L, k=5, sigma=-1e-6)                                                               # :106
        ev, evec = np.real(ev), np.real(evec)
        hist.append(ev.astype(np.float64))
        # the reference takes columns 0..2 of ARPACK's output, which for shift-invert are the eigenvalues
        # CLOSEST TO sigma first; the restatement selects them explicitly
        sel = np.argsort(np.abs(ev + 1e-6), kind="stable")[:3]
        v3 = evec[:, sel]
        r = v3 @ np.linalg.inv(v3[:3, :])                                                                  # :113
        r = polar_dual(r.reshape(N, 3, 3))[0].reshape(3 * N, 3).astype(r.dtype)                            # :115-116
        y = np.asarray(pr @ r).reshape(N, 3, 3)                                                            # :119
        u, sv, vt = np.linalg.svd(y)
        r = (u @ vt).reshape(3 * N, 3)                                                                     # :126-127 (no det fix)
        lam = (u * sv[:, None, :]) @ np.swapaxes(u, 1, 2)                                                  # :131
        lbd = csr_matrix((lam.reshape(-1), (rows, cols)), shape=(3 * N, 3 * N))                            # :133 (float64)
    if info is not None:
        info["evals"] = np.array(hist)
    return np.asarray(r, dtype=np.float64).reshape(N, 3, 3)


def bipartite_so3sync(src_edges, constraints, noise_model, edge_filter, maxiter, dtype=np.float32, info=None):
    """bipgo.py:18-142: {camera id: r, '<t>_0': r} with r the raw 3x3 blocks of the final iterate."""
    root = str(min(list(constraints.keys())))                                                              # :33
    r_0 = np.asarray(constraints[root].R())
    cam_s, time_s, blks, kr = [], [], [], []
    for key, val in src_edges.items():
        if not edge_filter(val):
            continue
        ts, mid = key[1].split("_")
        k_r = noise_model(val)
        blks.append(np.asarray(k_r * np.asarray(val["pose"].R()) @ np.asarray(constraints[mid].R()) @ r_0.T,
                               dtype=np.float64))                                                          # :44-45
        cam_s.append(key[0]); time_s.append(ts); kr.append(float(k_r))
    cam_nodes, ci = np.unique(np.char.add("c", np.array(cam_s, dtype=str)), return_inverse=True)           # :54
    time_nodes, ti = np.unique(np.char.add("t", np.array(time_s, dtype=str)), return_inverse=True)
    flat = dict(cam_names=cam_nodes, time_names=time_nodes, cam_idx=ci.astype(np.int64), time_idx=ti.astype(np.int64),
                wR=np.array(blks).reshape(-1, 3, 3), k_r=np.array(kr))
    mg = merge_edges(flat)                                                                                 # :47-52
    r = so3sync_general_arrays(mg["C"], mg["T"], mg["cam"], mg["time"], mg["blocks"], mg["a"], maxiter, dtype, info)
    out = {}
    for k, c in enumerate(cam_nodes):                                                                      # :135-141
        out[c[1:]] = r[k]
    for k, t in enumerate(time_nodes):
        out[t[1:] + "_0"] = r[mg["C"] + k]
    return out
