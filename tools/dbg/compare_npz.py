import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
bad = 0
for k in a.files:
    same = np.array_equal(a[k], b[k])
    print(k, "bit-identical" if same else "DIFFERENT max |d| %.3e" % np.abs(a[k] - b[k]).max())
    bad += not same
sys.exit(1 if bad else 0)
