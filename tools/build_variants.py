#!/usr/bin/env python3
"""Diagnostic builds of libvican_hip.so (in-tree, git-ignored, they travel to the GPU box):
    python tools/build_variants.py name=-DFLAG[,-DFLAG2] ...   ->  vican_amd/csrc/variants/libvican_hip_<name>.so
Use with VICAN_LIB=<path>."""
import os
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from vican_amd import _lib                                       # noqa: E402

out_dir = os.path.join(_lib.CSRC, "variants")
os.makedirs(out_dir, exist_ok=True)
for spec in sys.argv[1:]:
    name, _, flags = spec.partition("=")
    path = os.path.join(out_dir, "libvican_hip_%s.so" % name)
    _lib.build_library(out=path, extra_flags=tuple(f for f in flags.split(",") if f))
    print(path)
