import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gridpp_amd as gridpp
rng = np.random.default_rng(1)
f = rng.normal(0, 1, (4000, 4000)).astype(np.float32)
f[rng.random(f.shape) < 0.1] = np.nan
d = torch.from_numpy(f).cuda()
gridpp.fill_missing(d); torch.cuda.synchronize()
t0 = time.perf_counter(); out = gridpp.fill_missing(d); torch.cuda.synchronize(); print("fill_missing 4000^2 device: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
