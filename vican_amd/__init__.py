"""vican_amd -- MI355X-native bipartite SE(3) pose-graph solver.

Drop-in for the primal-dual path of the reference (`vican/bipgo.py:145-545`):
``bipartite_se3sync`` / ``object_bipartite_se3sync`` keep the edge-dict-in /
pose-dict-out API; the numerics run in hand-written HIP kernels for gfx950
behind the C ABI declared in ``include/vican_hip.h``.

Nothing in this package imports ``oracle/`` (test infrastructure only).
"""

__version__ = "0.1.0"

from .geometry import SE3  # noqa: F401
