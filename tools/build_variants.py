#!/usr/bin/env python3
"""Diagnostic builds of libvican_hip.so (in-tree, git-ignored, they travel to the GPU box):

    python tools/build_variants.py name=-DFLAG[,-DFLAG2] ...                     ->  vican_amd/csrc/variants/libvican_hip_<name>.so
    python tools/build_variants.py --patch tools/lab_patches/X.patch name=-DFLAG  ->  the same, built from a PATCHED COPY of the sources

Use with VICAN_LIB=<path>.  The shipped kernels carry no measurement apparatus (round 6: tools/strip_lab.py took the ablation
switches and phase stamps out of their bodies); the instrumented text lives as patches under tools/lab_patches/ (see its README
for the commit each one applies to).  --patch copies vican_amd/csrc and include/ to a scratch directory, applies the patch there
(`patch -p1`) and builds from the copy - the tree itself is never touched."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = __file__.rsplit("/", 2)[0]
sys.path.insert(0, ROOT)
from vican_amd import _lib                                       # noqa: E402

args = sys.argv[1:]
patch = None
if args[:1] == ["--patch"]:
    patch, args = os.path.abspath(args[1]), args[2:]
out_dir = os.path.join(_lib.CSRC, "variants")
os.makedirs(out_dir, exist_ok=True)
if patch:
    work = tempfile.mkdtemp(prefix="vican_lab_")
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(work, "include"))
    os.makedirs(os.path.join(work, "vican_amd"))
    shutil.copytree(_lib.CSRC, os.path.join(work, "vican_amd", "csrc"), ignore=shutil.ignore_patterns("*.so", "*.stamp", "variants", "_build*"))
    res = subprocess.run(["patch", "-p1", "-i", patch], cwd=work, capture_output=True, text=True)
    if res.returncode != 0:
        sys.exit("the patch does not apply to the current sources (see tools/lab_patches/README.md for its base commit):\n" + res.stdout + res.stderr)
    # point the build at the copy
    csrc, inc = os.path.join(work, "vican_amd", "csrc"), os.path.join(work, "include")
    rebase = lambda p: p.replace(_lib.CSRC, csrc).replace(_lib.INCLUDE, inc)
    _lib.SOURCES = [rebase(p) for p in _lib.SOURCES]
    extra = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hip") and os.path.join(csrc, f) not in _lib.SOURCES
             and f != "vican_wsweep.hip"]
    _lib.SOURCES += extra                                         # (translation units the patch brings back, e.g. vican_lres.hip)
    _lib.WSWEEP, _lib.HEADERS = rebase(_lib.WSWEEP), [rebase(p) for p in _lib.HEADERS]
    _lib.CSRC, _lib.INCLUDE = csrc, inc
for spec in args:
    name, _, flags = spec.partition("=")
    path = os.path.join(out_dir, "libvican_hip_%s.so" % name)
    _lib.build_library(out=path, extra_flags=tuple(f for f in flags.split(",") if f), force=bool(patch))
    print(path)
