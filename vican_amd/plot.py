"""2-D scatter view of a set of poses - the one plotting helper the reference notebook calls
(reference vican/plot.py:145-221, used by main.ipynb cell 11).  Outside the hot path; own, vectorised
implementation that only needs an object with a matplotlib-style ``scatter`` method."""
from __future__ import annotations

from typing import Iterable, Optional

import numpy as np

from .dataset import Camera
from .geometry import SE3

_VIEWS = {"xy": (0, 1), "xz": (0, 2), "yz": (1, 2)}


def plot2D(ax, data: dict, view: str, marker: str, s: float, c, invert: bool = False,
           idx: Optional[Iterable] = None, left_gauge: Optional[SE3] = None,
           right_gauge: Optional[SE3] = None) -> None:
    """Scatter the translations of ``left_gauge @ pose @ right_gauge`` (inverted afterwards if ``invert``)
    projected on the two axes named by ``view`` ('xy', 'xz' or 'yz').  ``data[n]`` is an SE3 or a Camera
    (its extrinsics are used); ``idx`` selects keys (default: all)."""
    if view not in _VIEWS:
        raise ValueError("view must be one of %s" % sorted(_VIEWS))
    keys = list(data.keys() if idx is None else idx)
    poses = np.empty((len(keys), 4, 4))
    for i, n in enumerate(keys):
        item = data[n]
        if isinstance(item, Camera):
            item = item.extrinsics
        elif not isinstance(item, SE3):
            raise TypeError("plot2D: data[%r] is neither a Camera nor an SE3" % (n,))
        poses[i, :3, :3], poses[i, :3, 3], poses[i, 3] = item.R(), item.t(), (0.0, 0.0, 0.0, 1.0)
    if left_gauge is not None:
        L = np.eye(4); L[:3, :3], L[:3, 3] = left_gauge.R(), left_gauge.t()
        poses = L[None] @ poses
    if right_gauge is not None:
        Rg = np.eye(4); Rg[:3, :3], Rg[:3, 3] = right_gauge.R(), right_gauge.t()
        poses = poses @ Rg[None]
    xyz = poses[:, :3, 3]
    if invert:                                              # translation of the inverse pose: -R^T t
        xyz = -np.einsum("nji,nj->ni", poses[:, :3, :3], xyz)
    a, b = _VIEWS[view]
    ax.scatter(xyz[:, a], xyz[:, b], s, marker=marker, c=c)
