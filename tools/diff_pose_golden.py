"""Largest differences between two pose-loop golden files.   python tools/diff_pose_golden.py new.npz [old.npz]"""
import sys, os
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
a = np.load(sys.argv[1])
b = np.load(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "pose_loop_golden.npz"))
for k in sorted(a.files):
    if k not in b.files:
        print("%-28s only in the new file" % k); continue
    x, y = np.asarray(a[k], np.float64), np.asarray(b[k], np.float64)
    if x.shape != y.shape:
        print("%-28s shapes %s vs %s" % (k, x.shape, y.shape)); continue
    m = np.isfinite(x) & np.isfinite(y)
    d = np.abs(x - y)[m]
    print("%-28s max |diff| %.3e   (max |value| %.3e, non-finite mismatch %d)" % (k, d.max() if d.size else 0.0, np.abs(y[m]).max() if d.size else 0.0,
                                                                                  int((np.isfinite(x) != np.isfinite(y)).sum())))
