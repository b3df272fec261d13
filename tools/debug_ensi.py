import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_ensi_parity import case, run
for kw in [dict(nan_member=2, nan_obs=True), dict(nan_member=None, nan_obs=False), dict(nan_member=2, nan_obs=False), dict(nan_member=None, nan_obs=True)]:
  for allow in (False, True):
    c = case(7, 20, 24, 8, 50, **kw)
    out, ref = run(c, 20000, 12, allow=allow)
    m = ~np.isnan(ref) & ~np.isnan(out)
    d = np.abs(out[m]-ref[m]); rel = d/np.maximum(np.abs(ref[m]),1e-2)
    print(kw, allow, "nan", np.isnan(out).sum(), np.isnan(ref).sum(), "max abs", d.max(), "max rel", rel.max(), "n bad", (rel>1e-5).sum())
    if (rel > 1e-5).any():
        idx = np.argwhere((np.abs(out-ref)/np.maximum(np.abs(ref),1e-2)) > 1e-5)[:6]
        for i in idx: print("   ", i, out[tuple(i)], ref[tuple(i)], c[2][tuple(i)])
