"""Rigid-transform container and SO(3)/SE(3) helpers.

Boundary type of the hot path: the reference pickles ``vican.geometry.SE3``
objects inside ``cam_marker_edges.pt`` and hands them in/out of
``bipartite_se3sync`` (reference geometry.py:194-261).  This is an independent
implementation keeping the same attribute names (``_R``, ``_t``, ``_pose``) so
reference pickles load, and the same dtype behaviour (``_pose`` is always
float32; ``inv()`` and ``@`` go through the float32 4x4, geometry.py:209,239-243,261)
because that rounding is visible in the solver's inputs.
"""
from __future__ import annotations

from typing import Iterable, Sequence

import numpy as np

__all__ = [
    "SE3", "project_SO3", "geodesic", "angle", "distance_SO3", "rad2deg",
    "deg2rad", "optimize_gauge_SO3", "optimize_gauge_SE3",
]


class SE3:
    """3-D rigid transform x -> R x + t.

    ``SE3(pose=P)`` keeps a float32 copy of the 4x4 and exposes views of it;
    ``SE3(R=R, t=t)`` keeps R and t as given (any dtype) plus a float32 4x4
    (same contract as reference geometry.py:195-218).
    """

    def __init__(self, **kw):
        if "pose" in kw:
            m = np.asarray(kw["pose"]).astype(np.float32)
            self._pose = m
            self._R = m[:3, :3]
            self._t = m[:3, 3]
            return
        rot = kw["R"]
        tr = np.asarray(kw["t"]).reshape(-1)
        m = np.zeros((4, 4), dtype=np.float32)
        m[:3, :3] = rot
        m[:3, 3] = tr
        m[3, 3] = 1.0
        self._R, self._t, self._pose = rot, tr, m

    # accessors -----------------------------------------------------------
    def R(self) -> np.ndarray:
        return self._R

    def t(self) -> np.ndarray:
        return self._t

    # algebra -------------------------------------------------------------
    def inv(self) -> "SE3":
        rt = self._R.T
        m = np.zeros((4, 4), dtype=np.float32)      # (= zeros_like(_pose): the 4x4 is always float32)
        m[:3, :3] = rt
        m[:3, 3] = -(rt @ self._t)
        m[3, 3] = 1
        out = SE3.__new__(SE3)                      # = SE3(pose=m) without its float32 copy of an array that is float32 already
        out._pose, out._R, out._t = m, m[:3, :3], m[:3, 3]
        return out

    def __matmul__(self, other: "SE3") -> "SE3":
        return SE3(pose=self._pose @ other._pose)

    def apply(self, pts: np.ndarray) -> np.ndarray:
        """Transform a 3 x n array of points."""
        if pts.ndim != 2 or pts.shape[0] != 3:
            raise AssertionError("apply() expects a 3 x n array")
        return self._R @ pts + self._t.reshape(3, 1)

    def __repr__(self) -> str:
        return str(np.round(self._pose, 4))


# ---------------------------------------------------------------------------
# SO(3) helpers
# ---------------------------------------------------------------------------

def project_SO3(x: np.ndarray) -> np.ndarray:
    """Nearest rotation (Frobenius) to a 3x3 matrix, det forced to +1
    (reference geometry.py:175-191)."""
    u, _, vh = np.linalg.svd(x)
    d = np.linalg.det(u @ vh)
    return (u * np.array([1.0, 1.0, d])) @ vh


def geodesic(ra: np.ndarray, rb: np.ndarray) -> np.ndarray:
    """Geodesic distance in RADIANS between rotations (broadcast over leading
    dims), using the chord form theta = 2 asin(|Ra-Rb|_F / (2 sqrt 2)).

    Unlike arccos((tr-1)/2) (reference geometry.py:149, noise floor ~1e-7 rad
    in f64 and ~5e-4 rad in f32) this is well conditioned near zero, so it can
    resolve the 1e-4 rad parity target.
    """
    d = np.asarray(ra, dtype=np.float64) - np.asarray(rb, dtype=np.float64)
    chord = np.sqrt(np.sum(d * d, axis=(-2, -1)))
    return 2.0 * np.arcsin(np.clip(chord / (2.0 * np.sqrt(2.0)), 0.0, 1.0))


def rad2deg(rad):
    return rad * 180.0 / np.pi


def deg2rad(deg):
    return deg * np.pi / 180.0


def angle(r: np.ndarray) -> float:
    """Rotation angle of a 3x3 rotation in DEGREES (notebook API,
    reference geometry.py:135-151), computed with the robust chord form."""
    return float(rad2deg(geodesic(r, np.eye(3))))


def distance_SO3(r1: np.ndarray, r2: np.ndarray) -> float:
    """Angle in DEGREES between two rotations (reference geometry.py:154-172)."""
    if r1.shape != (3, 3) or r2.shape != (3, 3):
        raise AssertionError("distance_SO3 expects two 3x3 matrices")
    return float(rad2deg(geodesic(r1, r2)))


def _procrustes_rotation(acc: np.ndarray) -> np.ndarray:
    u, _, vh = np.linalg.svd(acc.T)
    d = np.linalg.det(u @ vh)
    return (u * np.array([1.0, 1.0, d])) @ vh


def optimize_gauge_SO3(poses_a: Sequence[np.ndarray],
                       poses_b: Sequence[np.ndarray]) -> np.ndarray:
    """Rotation G minimising sum |A_i - B_i G|_F (reference geometry.py:264-291)."""
    if len(poses_a) != len(poses_b):
        raise AssertionError("length mismatch")
    acc = np.zeros((3, 3))
    for a, b in zip(poses_a, poses_b):
        acc += np.asarray(a, dtype=np.float64).T @ np.asarray(b, dtype=np.float64)
    return _procrustes_rotation(acc)


def optimize_gauge_SE3(poses_a: Iterable[SE3], poses_b: Iterable[SE3]) -> SE3:
    """Rigid gauge G aligning poses_b @ G with poses_a
    (reference geometry.py:294-325; used by main.ipynb cell 9)."""
    poses_a, poses_b = list(poses_a), list(poses_b)
    if len(poses_a) != len(poses_b):
        raise AssertionError("length mismatch")
    acc = np.zeros((3, 3))
    shift = np.zeros(3)
    for a, b in zip(poses_a, poses_b):
        rb = np.asarray(b.R(), dtype=np.float64)
        acc += np.asarray(a.R(), dtype=np.float64).T @ rb
        shift += rb.T @ (np.asarray(a.t(), dtype=np.float64) - np.asarray(b.t(), dtype=np.float64))
    return SE3(R=_procrustes_rotation(acc), t=shift / len(poses_a))
