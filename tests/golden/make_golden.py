#!/usr/bin/env python3
"""Generate golden parity vectors by RUNNING THE REAL REFERENCE (build container only).

    python tests/golden/make_golden.py            # writes tests/golden/<case>.npz

The reference (`/root/reference/vican`) is imported with an empty ``cv2`` stub
(its geometry.py:7 imports OpenCV only for ``langevin()``, which the solver never
calls).  Nothing of the reference is copied: the fixtures hold
only *data* - the flat input arrays of each case in ``tests/golden_cases.py`` and
the reference's outputs (pose dict as arrays, per-iteration eigenvalues from its
``eigs`` call, CG iteration count), plus the library versions used.

The reference never travels to the GPU box; tests read only the .npz files.
"""
import os
import sys
import types
import io
import contextlib

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

REF = "/root/reference"
if not os.path.isdir(REF):
    sys.exit("reference tree not present - goldens can only be generated in the build container")

sys.modules.setdefault("cv2", types.ModuleType("cv2"))       # see module docstring
sys.path.insert(0, REF)
import vican.bipgo as ref_bipgo          # noqa: E402  (the REAL reference)
import vican.geometry as ref_geometry    # noqa: E402
assert ref_bipgo.__file__.startswith(REF)

import golden_cases as gc                # noqa: E402
from vican_amd import synth              # noqa: E402

# -- instrumentation of the reference's third-party calls (recorded, not altered)
_rec = {"evals": [], "cg_iters": None}
_orig_eigs, _orig_cg = ref_bipgo.eigs, ref_bipgo.cg


def _eigs(*a, **k):
    ev, evec = _orig_eigs(*a, **k)
    _rec["evals"].append(np.real(ev).astype(np.float64))
    return ev, evec


def _cg(A, b, *a, **k):
    n = [0]

    keep = _rec.get("keep_iterates")
    early, last = {}, []

    def cb(_xk):
        n[0] += 1
        if keep is not None:                          # (scipy hands over its working array: copy)
            if n[0] in keep[0]:
                early[n[0]] = np.array(_xk, dtype=np.float64)
            last.append((n[0], np.array(_xk, dtype=np.float64)))
            del last[:-keep[1]]
    x, info = _orig_cg(A, b, *a, callback=cb, **k)
    _rec["cg_iters"] = n[0]
    if keep is not None:
        _rec["iterates"] = {**early, **dict(last)}
        _rec["x_final"] = np.array(x, dtype=np.float64)
        # the reference's OWN iterate-matched reproducibility: the same call on right-hand sides perturbed by one unit in
        # the last place, forced to run exactly as many iterations (no stopping test: rtol = 0) - how far ITERATE k moves
        rng = np.random.default_rng(12345)
        ks = sorted(_rec["iterates"])
        move = np.zeros((8, len(ks)))
        for trial in range(8):
            got, cnt = {}, [0]

            def cb2(_xk):
                cnt[0] += 1
                if cnt[0] in _rec["iterates"]:
                    got[cnt[0]] = np.array(_xk, dtype=np.float64)
            _orig_cg(A, b * (1.0 + 1e-15 * rng.standard_normal(b.shape)), *a, rtol=0.0, atol=0.0, maxiter=n[0], callback=cb2,
                     **{kk: vv for kk, vv in k.items() if kk not in ("rtol", "atol", "maxiter", "tol")})
            move[trial] = [float(np.linalg.norm((got[kk] - _rec["iterates"][kk]).reshape(-1, 3), axis=1).max()) for kk in ks]
        _rec["self_dx"] = move
    _rec["cg_relres"] = float(np.linalg.norm(b - A @ x) / np.linalg.norm(b))
    # the fully converged solution of the SAME system (x0 = 0 => same gauge): lets the tests
    # state how far the reference's loosely converged answer is from it
    xt, _ = _orig_cg(A, b, rtol=1e-14, maxiter=200000)
    _rec["t_tight"] = np.asarray(xt, dtype=np.float64).reshape(-1, 3)
    return x, info


ref_bipgo.eigs = _eigs
ref_bipgo.cg = _cg


def run_case(name, case):
    scene, flat = gc.build_flat(case)
    src = synth.edges_to_dict(flat, ref_geometry.SE3)
    out = {"name": name}
    for k in ("cam_key", "marker_key", "R", "t", "reprojected_err"):
        out["in_" + k] = flat[k]
    out["in_corners"] = flat["corners"].astype(np.float32)
    out["in_reprojected_err"] = flat["reprojected_err"].astype(np.float32)
    out["in_marker_ids"] = scene["marker_ids"]
    out["in_R_mk"] = scene["R_mk"]
    out["in_q_mk"] = scene["q_mk"]
    out["gt_R_cam"], out["gt_p_cam"] = scene["R_cam"], scene["p_cam"]
    out["gt_R_obj"], out["gt_p_obj"] = scene["R_obj"], scene["p_obj"]
    nr, nt, ff = (gc.CALLABLES[case[k]] for k in ("noise_r", "noise_t", "filt"))
    for solver, dt in case["runs"]:
        _rec["evals"], _rec["cg_iters"], _rec["cg_relres"], _rec["t_tight"] = [], None, None, None
        dtype = np.dtype(dt).type
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            if case["mode"] == "camera":
                cons = synth.constraints_from_scene(scene, ref_geometry.SE3)
                res = ref_bipgo.bipartite_se3sync(src, constraints=cons, noise_model_r=nr,
                                                  noise_model_t=nt, edge_filter=ff,
                                                  maxiter=gc.MAXITER, lsqr_solver=solver, dtype=dtype)
            else:
                res = ref_bipgo.object_bipartite_se3sync(src, noise_model_r=nr, noise_model_t=nt,
                                                         edge_filter=ff, maxiter=gc.MAXITER,
                                                         lsqr_solver=solver, dtype=dtype)
        tag = "out_%s_%s_" % (solver, dt)
        keys = list(res.keys())
        out[tag + "keys"] = np.array([str(k) for k in keys])
        out[tag + "R"] = np.stack([np.asarray(res[k].R(), dtype=np.float64) for k in keys])
        out[tag + "t"] = np.stack([np.asarray(res[k].t(), dtype=np.float64) for k in keys])
        out[tag + "evals"] = np.stack(_rec["evals"]) if _rec["evals"] else np.zeros((0, 5))
        out[tag + "cg_iters"] = np.int64(-1 if _rec["cg_iters"] is None else _rec["cg_iters"])
        out[tag + "cg_relres"] = np.float64(np.nan if _rec["cg_relres"] is None else _rec["cg_relres"])
        if _rec.get("t_tight") is not None:
            tt = _rec["t_tight"]
            out[tag + "t_tight"] = tt[:len(keys)] if case["mode"] == "camera" else np.zeros((0, 3))
            out[tag + "dist_tight"] = np.float64(np.linalg.norm(
                np.stack([np.asarray(v.t(), dtype=np.float64) for v in res.values()]) - tt[:len(keys)], axis=1).max()
                if case["mode"] == "camera" else np.nan)
        print("  %-12s %-20s %-8s nodes=%d cg_iters=%s evals[-1]=%s" % (
            name, solver, dt, len(keys), _rec["cg_iters"],
            np.array2string(out[tag + "evals"][-1], precision=3) if len(_rec["evals"]) else "-"))
    out["versions"] = np.array(["numpy " + np.__version__, "scipy " + scipy.__version__,
                                "python " + sys.version.split()[0]])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def polar_case():
    """G6: inputs for the batched 3x3 polar/dual kernel incl. reflections and
    near-rank-deficient blocks, with the reference's own project_SO3 outputs."""
    rng = np.random.default_rng(61)
    n = 256
    x = rng.standard_normal((n, 3, 3))
    rots = synth.random_rotations(rng, n)
    x[:64] = rots[:64] * rng.uniform(0.5, 20.0, (64, 1, 1))                 # scaled rotations
    # reflections (det < 0) with DISTINCT singular values: the nearest rotation of a scaled
    # reflection c*Q is not unique, so equal singular values cannot be a parity vector
    x[64:96] = rots[64:96] @ np.diag([3.0, 2.0, -1.0]) @ np.swapaxes(rots[96:128], 1, 2)
    u, v = synth.random_rotations(rng, 32), synth.random_rotations(rng, 32)
    s = np.stack([np.array([5.0, 1.0, 1e-9 * (i + 1)]) for i in range(32)])
    x[96:128] = (u * s[:, None, :]) @ v                                       # near rank-2
    x[128:144] = (u[:16] * np.array([3.0, 3.0, 1.0])[None, None, :]) @ v[:16]  # repeated sigma
    proj = np.stack([ref_geometry.project_SO3(m) for m in x])
    np.savez_compressed(os.path.join(HERE, "g6_polar.npz"), x=x, project_SO3=proj)
    print("  g6_polar     %d blocks" % n)


def pickle_case():
    """A small `cam_marker_edges.pt`-style file written with the REFERENCE's SE3 class (torch.save of
    the edge dict, main.ipynb:68): the fixture is pickled data that names `vican.geometry.SE3`; the
    test loads it through this repository's shim to prove reference caches unpickle unchanged."""
    import torch
    scene, flat = gc.build_flat(gc.CASES["g5_strings"])
    src = synth.edges_to_dict(flat, ref_geometry.SE3)
    some = dict(list(src.items())[:6])
    # one pose built from a 4x4 (float32 views) as cam.py would after SE3(pose=...), one inverted
    k0 = next(iter(some))
    some[k0]["pose"] = ref_geometry.SE3(pose=np.eye(4) + 0.01 * np.arange(16).reshape(4, 4))
    k1 = list(some)[1]
    some[k1]["pose"] = some[k1]["pose"].inv()
    torch.save(some, os.path.join(HERE, "ref_edges_pickle.pt"))
    np.savez_compressed(os.path.join(HERE, "ref_edges_pickle_expect.npz"),
                        keys=np.array(["|".join(k) for k in some]),
                        R=np.stack([np.asarray(v["pose"].R(), dtype=np.float64) for v in some.values()]),
                        t=np.stack([np.asarray(v["pose"].t(), dtype=np.float64) for v in some.values()]),
                        R_dtype=np.array([str(np.asarray(v["pose"].R()).dtype) for v in some.values()]))
    print("  ref_edges_pickle.pt  %d edges" % len(some))


def eval_case():
    """G7: dataset reader + evaluation harness.  A small cameras.json / object_pose file pair (own synthetic
    data) is parsed by the REFERENCE's Dataset.read_cameras / read_object, and the notebook's cell-9 error
    statistics are computed with the REFERENCE's optimize_gauge_SE3 / distance_SO3 / angle for a perturbed,
    gauge-shifted estimate of the cameras.  The fixture holds the JSON text, the estimate and those outputs."""
    import json
    import tempfile
    import vican.dataset as ref_dataset
    rng = np.random.default_rng(77)
    ids = ["0", "3", "10", "11", "25", "100", "7"]
    R = synth.random_rotations(rng, len(ids)); t = rng.normal(0, 5.0, (len(ids), 3))
    cams = {}
    for i, c in enumerate(ids):
        cams[c] = dict(fx=1000.0 + i, fy=990.0 - i, cx=640.5, cy=360.25, distortion=[0.01 * i] * 12, R=R[i].tolist(),
                       t=t[i].tolist(), resolution_x=1280, resolution_y=720)
    obj = {str(k): dict(R=synth.random_rotations(rng, 1)[0].tolist(), t=rng.normal(0, 1, 3).tolist()) for k in range(5)}
    cam_text, obj_text = json.dumps(cams), json.dumps(obj)
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "cameras.json"), "w").write(cam_text)
        open(os.path.join(d, "object_pose_0.json"), "w").write(obj_text)
        ds = ref_dataset.Dataset(root=d)
    # estimate: ground truth moved by a global gauge, perturbed; camera "7" missing, one extra id
    Gr = synth.random_rotations(rng, 1)[0]; Gt = rng.normal(0, 2.0, 3)
    est = {}
    for i, c in enumerate(ids[:-1]):
        dR = synth.small_rotations(rng, 1, 0.02)[0] if hasattr(synth, "small_rotations") else ref_geometry.project_SO3(np.eye(3) + 0.02 * rng.standard_normal((3, 3)))
        est[c] = ref_geometry.SE3(R=Gr @ R[i] @ dR, t=Gr @ t[i] + Gt + 0.03 * rng.standard_normal(3))
    est["999"] = ref_geometry.SE3(R=np.eye(3), t=np.zeros(3))
    valid = [c for c in ds.cams.keys() if c in est]
    G = ref_geometry.optimize_gauge_SE3([ds.cams[c].extrinsics.inv() for c in valid], [est[c].inv() for c in valid])
    r_err, t_err, xyz = [], [], []
    for c in valid:
        gt = ds.cams[c].extrinsics
        e = G.inv() @ est[c]
        t_err.append(np.linalg.norm(gt.t() - e.t(), ord=2) * 100)
        r_err.append(ref_geometry.distance_SO3(gt.R(), e.R()))
        xyz.append(np.abs(gt.t() - e.t()) * 100)
    np.savez_compressed(os.path.join(HERE, "g7_eval.npz"), cameras_json=np.array(cam_text), object_json=np.array(obj_text),
                        cam_ids=np.array(list(ds.cams.keys())),
                        K=np.stack([ds.cams[c].intrinsics for c in ds.cams]), dist=np.stack([ds.cams[c].distortion for c in ds.cams]),
                        ext_R=np.stack([ds.cams[c].extrinsics.R() for c in ds.cams]), ext_t=np.stack([ds.cams[c].extrinsics.t() for c in ds.cams]),
                        obj_keys=np.array(list(ds.object.keys())), obj_R=np.stack([v.R() for v in ds.object.values()]),
                        obj_t=np.stack([v.t() for v in ds.object.values()]),
                        est_ids=np.array(list(est.keys())), est_R=np.stack([v.R() for v in est.values()]),
                        est_t=np.stack([v.t() for v in est.values()]),
                        valid=np.array(valid), gauge_R=np.asarray(G.R(), dtype=np.float64), gauge_t=np.asarray(G.t(), dtype=np.float64),
                        r_err=np.array(r_err), t_err=np.array(t_err), xyz_err=np.stack(xyz),
                        angle_deg=np.array([ref_geometry.angle(np.asarray(v.R())) for v in est.values()]))
    # real-capture layout (DojoDataset, dataset.py:103-181), parsed by the reference
    pose = lambda: np.block([[synth.random_rotations(rng, 1)[0], rng.normal(0, 2, (3, 1))], [np.zeros((1, 3)), np.ones((1, 1))]])
    dj_intr = {c: dict(intrinsics=(np.eye(3) * (900 + i) + np.array([[0, 0, 640.0], [0, 0, 360.0], [0, 0, 0]])).tolist(),
                       distortion=(0.001 * (i + 1) * np.arange(5)).tolist()) for i, c in enumerate(ids[:4])}
    dj_extr = {c: pose().tolist() for c in ids[:4]}
    dj_cube = {"to": {str(m): pose().tolist() for m in range(6)}}
    with tempfile.TemporaryDirectory() as d:
        json.dump(dj_intr, open(os.path.join(d, "cameras_intrinsics.json"), "w"))
        json.dump(dj_extr, open(os.path.join(d, "cameras_transformations_to_origin_ground_truth.json"), "w"))
        json.dump(dj_cube, open(os.path.join(d, "aruco_cube_transformations.json"), "w"))
        os.makedirs(os.path.join(d, "aruco_images_samples", "5"))
        open(os.path.join(d, "aruco_images_samples", "5", ids[1] + ".jpg"), "w").close()
        dj = ref_dataset.DojoDataset(root=d)
    np.savez_compressed(os.path.join(HERE, "g7_dojo.npz"), intr_json=np.array(json.dumps(dj_intr)), extr_json=np.array(json.dumps(dj_extr)),
                        cube_json=np.array(json.dumps(dj_cube)), cam_ids=np.array(list(dj.cams.keys())),
                        K=np.stack([dj.cams[c].intrinsics for c in dj.cams]), dist=np.stack([dj.cams[c].distortion for c in dj.cams]),
                        ext_R=np.stack([dj.cams[c].extrinsics.R() for c in dj.cams]), ext_t=np.stack([dj.cams[c].extrinsics.t() for c in dj.cams]),
                        con_ids=np.array(list(dj.object_constraints.keys())),
                        con_R=np.stack([v.R() for v in dj.object_constraints.values()]),
                        con_t=np.stack([v.t() for v in dj.object_constraints.values()]),
                        im_cam_id=np.array(dj.im_data["cam_id"]), im_timestamp=np.array(dj.im_data["timestamp"]))
    print("  g7_eval      %d cameras (+ g7_dojo)" % len(ids))


SO3_CASES = ("g2_small", "g3_medium", "g5_strings")


def so3_case():
    """G8: the non-eliminated variant `bipartite_so3sync` (bipgo.py:18-142) on the inputs of the camera
    cases above (their fixtures hold the inputs); outputs only: rotation dict + the eigs values."""
    out = {}
    for name in SO3_CASES:
        case = gc.CASES[name]
        scene, flat = gc.build_flat(case)
        src = synth.edges_to_dict(flat, ref_geometry.SE3)
        cons = synth.constraints_from_scene(scene, ref_geometry.SE3)
        nr, ff = gc.CALLABLES[case["noise_r"]], gc.CALLABLES[case["filt"]]
        for dt in ("float64", "float32"):
            _rec["evals"] = []
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                res = ref_bipgo.bipartite_so3sync(src, constraints=cons, noise_model=nr, edge_filter=ff,
                                                  maxiter=gc.MAXITER, dtype=np.dtype(dt).type)
            tag = "%s_%s_" % (name, dt)
            keys = list(res.keys())
            out[tag + "keys"] = np.array([str(k) for k in keys])
            out[tag + "R"] = np.stack([np.asarray(res[k], dtype=np.float64) for k in keys])
            out[tag + "evals"] = np.stack(_rec["evals"])
            print("  g8_so3sync   %-12s %-8s nodes=%d evals[-1]=%s" % (
                name, dt, len(keys), np.array2string(out[tag + "evals"][-1], precision=3)))
    out["versions"] = np.array(["numpy " + np.__version__, "scipy " + scipy.__version__,
                                "python " + sys.version.split()[0]])
    np.savez_compressed(os.path.join(HERE, "g8_so3sync.npz"), **out)


def input_digest(flat):
    """Order-sensitive digest of the generated source edges (so a test knows it regenerated the same inputs)."""
    return np.array([float(np.sum(flat["R"] * np.arange(1, flat["R"].size + 1).reshape(flat["R"].shape) % 7)),
                     float(np.sum(flat["t"])), float(np.sum(flat["corners"])), float(len(flat["cam_key"]))])


def large_shop_case():
    """G9: the reference at BASELINE configs[2] scale (see golden_cases.LARGE_SHOP); outputs + input digest only."""
    import time
    case = gc.LARGE_SHOP
    scene, flat = gc.build_flat(case)
    src = synth.edges_to_dict(flat, ref_geometry.SE3)
    cons = synth.constraints_from_scene(scene, ref_geometry.SE3)
    nr, nt, ff = (gc.CALLABLES[case[k]] for k in ("noise_r", "noise_t", "filt"))
    out = {"digest": input_digest(flat)}
    for solver, dt in case["runs"]:
        _rec["evals"], _rec["cg_iters"], _rec["cg_relres"], _rec["t_tight"] = [], None, None, None
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            res = ref_bipgo.bipartite_se3sync(src, constraints=cons, noise_model_r=nr, noise_model_t=nt, edge_filter=ff,
                                              maxiter=gc.MAXITER, lsqr_solver=solver, dtype=np.dtype(dt).type)
        wall = time.time() - t0
        tag = "out_%s_%s_" % (solver, dt)
        keys = list(res.keys())
        out[tag + "keys"] = np.array([str(k) for k in keys])
        Rs = np.stack([np.asarray(res[k].R(), dtype=np.float64) for k in keys])
        out[tag + "R"] = Rs.astype(np.float32) if dt == "float32" else Rs              # f32 run: float32 holds it exactly
        out[tag + "t"] = np.stack([np.asarray(res[k].t(), dtype=np.float64) for k in keys])
        out[tag + "evals"] = np.stack(_rec["evals"])
        out[tag + "cg_iters"] = np.int64(_rec["cg_iters"])
        out[tag + "cg_relres"] = np.float64(_rec["cg_relres"])
        tt = _rec["t_tight"]
        out[tag + "dist_tight"] = np.float64(np.linalg.norm(out[tag + "t"] - tt[:len(keys)], axis=1).max())
        out[tag + "ref_wall_s"] = np.float64(wall)
        print("  g9_large_shop %-20s %-8s nodes=%d src_edges=%d cg_iters=%s dist_tight=%.3e reference wall %.1f s" % (
            solver, dt, len(keys), len(src), _rec["cg_iters"], out[tag + "dist_tight"], wall))
    out["versions"] = np.array(["numpy " + np.__version__, "scipy " + scipy.__version__, "python " + sys.version.split()[0]])
    np.savez_compressed(os.path.join(HERE, "g9_large_shop.npz"), **out)


def _run_reference(case, solver, dt):
    scene, flat = gc.build_flat(case)
    src = synth.edges_to_dict(flat, ref_geometry.SE3)
    cons = synth.constraints_from_scene(scene, ref_geometry.SE3)
    nr, nt, ff = (gc.CALLABLES[case[k]] for k in ("noise_r", "noise_t", "filt"))
    import time
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        res = ref_bipgo.bipartite_se3sync(src, constraints=cons, noise_model_r=nr, noise_model_t=nt, edge_filter=ff,
                                          maxiter=gc.MAXITER, lsqr_solver=solver, dtype=np.dtype(dt).type)
    return flat, res, time.time() - t0


def cg_iterates_case():
    """G10: the reference's CG iterates x_k (scipy's callback on its own cg call, bipgo.py:477) for the k in
    golden_cases.ITERATE_EARLY and the last ITERATE_LAST iterations of each run, stored as float32 differences to the
    run's final iterate (which is stored in float64, like the rotations that produced the right-hand side), in the order
    of the reference's output dict (= its node order).  One reference run per case and dtype: everything in a tag is
    consistent with everything else in it."""
    out = {}
    for name, case in gc.ITERATE_CASES.items():
        for solver, dt in case["runs"]:
            if solver != "conjugate_gradient":
                continue
            _rec.update(evals=[], cg_iters=None, keep_iterates=(set(gc.ITERATE_EARLY), gc.ITERATE_LAST))
            flat, res, wall = _run_reference(case, solver, dt)
            _rec["keep_iterates"] = None
            tag = "%s_%s_" % (name, dt)
            keys = list(res.keys())
            t = np.stack([np.asarray(res[k].t(), dtype=np.float64) for k in keys])
            assert np.array_equal(t.reshape(-1), _rec["x_final"]), "output translations are the CG vector in node order"
            Rs = np.stack([np.asarray(res[k].R(), dtype=np.float64) for k in keys])
            ks = sorted(_rec["iterates"])
            out[tag + "digest"] = input_digest(flat)
            out[tag + "keys"] = np.array([str(k) for k in keys])
            out[tag + "R"] = Rs.astype(np.float32) if dt == "float32" else Rs
            out[tag + "t"] = t
            out[tag + "cg_iters"] = np.int64(_rec["cg_iters"])
            out[tag + "k"] = np.array(ks, dtype=np.int64)
            out[tag + "dx"] = np.stack([(_rec["iterates"][k] - _rec["x_final"]).reshape(-1, 3) for k in ks]).astype(np.float32)
            out[tag + "self_dx"] = _rec["self_dx"]              # [8 trials, len(k)]: movement of the reference's own iterate k
            step = [float(np.linalg.norm((_rec["iterates"][b] - _rec["iterates"][a]).reshape(-1, 3), axis=1).max())
                    for a, b in zip(ks[-gc.ITERATE_LAST:-1], ks[-gc.ITERATE_LAST + 1:])]
            print("  g10_iterates %-14s %-8s cg_iters=%d kept k=%s  last steps move %s m  (%.1f s)" % (
                name, dt, _rec["cg_iters"], ks, " ".join("%.1e" % v for v in step), wall))
            print("      iterate-matched self-movement under 1e-15 perturbations (max of 8): %s" % " ".join(
                "%d:%.1e" % (k, v) for k, v in zip(ks, _rec["self_dx"].max(0))))
    out["versions"] = np.array(["numpy " + np.__version__, "scipy " + scipy.__version__, "python " + sys.version.split()[0]])
    np.savez_compressed(os.path.join(HERE, "g10_cg_iterates.npz"), **out)


def unit_scale_case():
    """G11: the reference on UNIT-weight scenes of large_shop and small_room size (golden_cases.UNIT_SCALE)."""
    out = {}
    for name, case in gc.UNIT_SCALE.items():
        for solver, dt in case["runs"]:
            _rec.update(evals=[], cg_iters=None, cg_relres=None, t_tight=None)
            flat, res, wall = _run_reference(case, solver, dt)
            tag = "%s_%s_" % (name, dt)
            keys = list(res.keys())
            Rs = np.stack([np.asarray(res[k].R(), dtype=np.float64) for k in keys])
            out[name + "_digest"] = input_digest(flat)
            out[tag + "keys"] = np.array([str(k) for k in keys])
            out[tag + "R"] = Rs.astype(np.float32) if dt == "float32" else Rs
            out[tag + "t"] = np.stack([np.asarray(res[k].t(), dtype=np.float64) for k in keys])
            out[tag + "evals"] = np.stack(_rec["evals"])
            out[tag + "cg_iters"] = np.int64(_rec["cg_iters"])
            out[tag + "dist_tight"] = np.float64(np.linalg.norm(out[tag + "t"] - _rec["t_tight"][:len(keys)], axis=1).max())
            out[tag + "ref_wall_s"] = np.float64(wall)
            print("  g11_unit     %-22s %-8s nodes=%d src_edges=%d cg_iters=%d dist_tight=%.2e m  reference wall %.1f s" % (
                name, dt, len(keys), len(flat["cam_key"]), _rec["cg_iters"], out[tag + "dist_tight"], wall))
    out["versions"] = np.array(["numpy " + np.__version__, "scipy " + scipy.__version__, "python " + sys.version.split()[0]])
    np.savez_compressed(os.path.join(HERE, "g11_unit_scale.npz"), **out)


def cube_calib_case():
    """G12: the reference's object calibration at the notebook's size and weights (golden_cases.CUBE_CALIB; main.ipynb:74-80).
    Besides the returned marker poses: the eigenvalues per iteration, the CG iteration count, the converged solution of the
    reference's own system for the marker nodes (`t_tight`: its node order comes from the inner bipartite_se3sync call, whose
    full output dict is observed, not altered) and the reference's own reproducibility - eight repeats of its cg call on
    right-hand sides perturbed by one unit in the last place (movement of the MARKER translations, iteration counts)."""
    import time
    case = gc.CUBE_CALIB
    scene, flat = gc.build_flat(case)
    src = synth.edges_to_dict(flat, ref_geometry.SE3)
    nr, nt, ff = (gc.CALLABLES[case[k]] for k in ("noise_r", "noise_t", "filt"))
    out = {"digest": input_digest(flat)}
    inner, seen = ref_bipgo.bipartite_se3sync, {}
    cg_inner = ref_bipgo.cg                                     # (= _cg above: records iterations, relres, t_tight)

    def spy(*a, **k):
        res = inner(*a, **k)
        seen["keys"] = [str(x) for x in res.keys()]
        return res

    def cg_spy(A, b, *a, **k):
        x, info = cg_inner(A, b, *a, **k)
        rng = np.random.default_rng(12345)
        xs, its = [], []
        for _ in range(8):
            n = [0]
            xt, _ = _orig_cg(A, b * (1.0 + 1e-15 * rng.standard_normal(b.shape)), *a, callback=lambda _x: n.__setitem__(0, n[0] + 1), **k)
            xs.append(np.asarray(xt, dtype=np.float64).reshape(-1, 3)); its.append(n[0])
        seen["trials"], seen["trial_iters"], seen["x"] = xs, its, np.asarray(x, dtype=np.float64).reshape(-1, 3)
        return x, info
    ref_bipgo.bipartite_se3sync, ref_bipgo.cg = spy, cg_spy
    try:
        for solver, dt in case["runs"]:
            _rec.update(evals=[], cg_iters=None, cg_relres=None, t_tight=None)
            t0 = time.time()
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                res = ref_bipgo.object_bipartite_se3sync(src, noise_model_r=nr, noise_model_t=nt, edge_filter=ff, maxiter=gc.MAXITER,
                                                         lsqr_solver=solver, dtype=np.dtype(dt).type)
            wall = time.time() - t0
            tag = "out_%s_%s_" % (solver, dt)
            keys = [str(k) for k in res.keys()]
            rows = np.array([seen["keys"].index(k) for k in keys])          # marker nodes inside the inner call's node order
            t = np.stack([np.asarray(res[k].t(), dtype=np.float64) for k in res.keys()])
            assert np.array_equal(seen["x"][rows], t)
            out[tag + "keys"] = np.array(keys)
            out[tag + "R"] = np.stack([np.asarray(res[k].R(), dtype=np.float64) for k in res.keys()])
            out[tag + "t"] = t
            out[tag + "evals"] = np.stack(_rec["evals"])
            out[tag + "cg_iters"] = np.int64(_rec["cg_iters"])
            out[tag + "cg_relres"] = np.float64(_rec["cg_relres"])
            out[tag + "t_tight"] = _rec["t_tight"][rows]
            out[tag + "dist_tight"] = np.float64(np.linalg.norm(t - _rec["t_tight"][rows], axis=1).max())
            out[tag + "self_move"] = np.array([float(np.linalg.norm(xt[rows] - t, axis=1).max()) for xt in seen["trials"]])
            out[tag + "self_move_all_nodes"] = np.array([float(np.linalg.norm(xt - seen["x"], axis=1).max()) for xt in seen["trials"]])
            out[tag + "iters"] = np.array([_rec["cg_iters"]] + seen["trial_iters"], dtype=np.int64)
            out[tag + "n_nodes"] = np.int64(len(seen["keys"]))
            out[tag + "ref_wall_s"] = np.float64(wall)
            print("  g12_cube_calib %-20s %-8s markers=%d nodes=%d src_edges=%d cg_iters=%s (trials %s) relres %.1e dist_tight=%.3e m "
                  "self-movement of the marker translations %s m  reference wall %.1f s" % (
                      solver, dt, len(keys), len(seen["keys"]), len(src), _rec["cg_iters"], seen["trial_iters"], _rec["cg_relres"], out[tag + "dist_tight"],
                      " ".join("%.1e" % v for v in out[tag + "self_move"]), wall))
    finally:
        ref_bipgo.bipartite_se3sync, ref_bipgo.cg = inner, cg_inner
    out["versions"] = np.array(["numpy " + np.__version__, "scipy " + scipy.__version__, "python " + sys.version.split()[0]])
    np.savez_compressed(os.path.join(HERE, "g12_cube_calib.npz"), **out)


if __name__ == "__main__":
    only = sys.argv[1:]
    for name, case in gc.CASES.items():
        if only and name not in only:
            continue
        run_case(name, case)
    if not only or "g6_polar" in only:
        polar_case()
    if not only or "pickle" in only:
        pickle_case()
    if not only or "g7_eval" in only:
        eval_case()
    if not only or "g8_so3sync" in only:
        so3_case()
    if "g9_large_shop" in only:               # minutes of reference time: only on request
        large_shop_case()
    if "g10_cg_iterates" in only:
        cg_iterates_case()
    if "g11_unit_scale" in only:
        unit_scale_case()
    if "g12_cube_calib" in only:
        cube_calib_case()
