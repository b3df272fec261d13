import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_oi_parity import make_case, run_both
for mp in (5, 20):
    c = make_case(1001, 64, 64, 200)
    out, ref = run_both(c, 10000, 0, 0, mp)
    d = np.abs(out.astype(np.float64) - ref)
    rel = d / np.maximum(np.abs(ref), 1e-3)
    idx = np.argsort(rel.ravel())[::-1][:8]
    print("max_points", mp, "max abs", d.max(), "max rel", rel.max(), "n>1e-6:", (rel > 1e-6).sum())
    for i in idx:
        y, x = divmod(i, 64)
        print("  cell", y, x, "out %.9g ref %.9g bg %.9g abs %.3g" % (out[y, x], ref[y, x], c["bg"][y, x], d[y, x]))
