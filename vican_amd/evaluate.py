"""Evaluation harness: calibration error against ground truth exactly as the
reference notebook reports it (main.ipynb cell 9): rigid gauge alignment with
``optimize_gauge_SE3`` on the inverted poses, then per-camera rotation error in degrees
(``distance_SO3``) and translation errors in centimetres (norm and per axis)."""
from __future__ import annotations

import numpy as np

from .geometry import distance_SO3, optimize_gauge_SE3

__all__ = ["calibration_errors", "format_error_table"]

_STATS = (("min", np.min), ("avg", np.mean), ("std", np.std), ("median", np.median), ("max", np.max))


def calibration_errors(gt: dict, pose_est: dict) -> dict:
    """``gt``: ``{camera id: SE3}`` ground-truth extrinsics (or ``Dataset.cams``: objects with
    ``.extrinsics``); ``pose_est``: output of ``bipartite_se3sync``.  Returns the raw per-camera
    error arrays, the missing / valid id lists, the gauge and the min/avg/std/median/max table."""
    ext = {c: (v.extrinsics if hasattr(v, "extrinsics") else v) for c, v in gt.items()}
    missing = [c for c in ext if c not in pose_est]
    valid = [c for c in ext if c in pose_est]
    if not valid:
        raise ValueError("no camera of the ground truth is present in the estimate")
    G = optimize_gauge_SE3([ext[c].inv() for c in valid], [pose_est[c].inv() for c in valid])
    Gi = G.inv()
    r_err, t_err, axis = [], [], ([], [], [])
    for c in valid:
        est = Gi @ pose_est[c]
        d = np.asarray(ext[c].t(), dtype=np.float64) - np.asarray(est.t(), dtype=np.float64)
        t_err.append(np.linalg.norm(d, ord=2) * 100.0)
        r_err.append(distance_SO3(np.asarray(ext[c].R()), np.asarray(est.R())))
        for k in range(3):
            axis[k].append(abs(d[k]) * 100.0)
    rows = {"SO(3)": np.array(r_err), "E(3)": np.array(t_err), "X": np.array(axis[0]), "Y": np.array(axis[1]),
            "Z": np.array(axis[2])}
    table = {name: {s: float(fn(v)) for s, fn in _STATS} for name, v in rows.items()}
    return {"missing": missing, "valid": valid, "gauge": G, "errors": rows, "table": table}


def format_error_table(res: dict) -> str:
    """Text of the notebook's cell 9 (degrees for SO(3), centimetres otherwise)."""
    lines = ["Missing cameras: {}".format(res["missing"] if res["missing"] else "None")]
    for name, st in res["table"].items():
        u = "deg" if name == "SO(3)" else "cm"
        lines.append("{}\t min: {:.3f}{u} | avg: {:.3f}{u} | std: {:.3f}{u} | median: {:.3f}{u} | max: {:.3f}{u}".format(
            name, st["min"], st["avg"], st["std"], st["median"], st["max"], u=u))
    return "\n".join(lines)
