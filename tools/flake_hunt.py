#!/usr/bin/env python3
"""Repeat one parametrised GPU kernel test in-process and report every failure (hunting rare nondeterminism)."""
import sys, traceback
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests"); sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
import test_kernels_gpu as t
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
bad = 0
for i in range(n):
    for cfg in (t.CONFIGS[5], t.CONFIGS[4], t.CONFIGS[6]):
        for dt in (np.float32, np.float64):
            try:
                t.test_block_op_dual_update_and_init(cfg, dt)
            except AssertionError:
                bad += 1
                tb = traceback.format_exc().splitlines()
                print("FAIL rep %d cfg %s dt %s: %s" % (i, cfg, dt.__name__, [l for l in tb if "assert" in l][:2]))
print("done: %d failures in %d repetitions" % (bad, n))
