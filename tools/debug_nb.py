import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gridpp_amd as gridpp
from oracle import oracle as O
from tests.test_gpu_neighbourhood_parity import field
f = field(1, 257, 193)
for stat in (gridpp.Std, gridpp.Variance):
    out = gridpp.neighbourhood(f, 0, stat); ref = O.neighbourhood(f, 0, stat)
    m = ~np.isnan(ref) & ~np.isnan(out)
    print(stat, "nan out", np.isnan(out).sum(), "nan ref", np.isnan(ref).sum(), "m", m.sum(), "maxabs", np.abs(out[m]-ref[m]).max() if m.sum() else None, "eq", (out[m]==ref[m]).mean() if m.sum() else None)
