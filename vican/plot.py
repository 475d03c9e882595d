"""Import-path shim for ``from vican.plot import plot2D`` (main.ipynb cell 1)."""
from vican_amd.plot import plot2D  # noqa: F401
