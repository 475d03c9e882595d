from vican_amd.bipgo import bipartite_se3sync, bipartite_so3sync, object_bipartite_se3sync  # noqa: F401
