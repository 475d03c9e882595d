"""Import-path shim: ``vican.bipgo`` / ``vican.geometry`` resolve to the MI355X build
(``vican_amd``) so the upstream ``main.ipynb`` and pickled ``cam_marker_edges.pt`` files
(which reference ``vican.geometry.SE3``) work unchanged."""
