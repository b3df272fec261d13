"""The headline OI call timed like bench.py does (inputs resident in HBM, K steps between synchronisations): ms per step, the first pass's
HIP-event time and all OI kernels' -- for A/B runs of library variants (GPP_LIB), which bench.py refuses.  usage: oi_time.py [steps] [ny] [obs] [max_points]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gridpp_amd as gridpp
from tools.bench_cases import make_workload
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ny = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
S = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
mp = int(sys.argv[4]) if len(sys.argv) > 4 else 30
lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(ny, ny, S, 1002, 0, ny)
grid, points, st = gridpp.Grid(lats, lons), gridpp.Points(plat, plon), gridpp.BarnesStructure(10000)
d = [torch.from_numpy(a).cuda() for a in (bg, obs, ratios, pbg)]
for _ in range(3):
    gridpp.optimal_interpolation(grid, d[0], points, d[1], d[2], d[3], st, mp)
torch.cuda.synchronize()
u, k = [], []
t0 = time.perf_counter()
for _ in range(steps):
    gridpp.optimal_interpolation(grid, d[0], points, d[1], d[2], d[3], st, mp)
    s = gridpp.oi_last_stats(); u.append(s["union_kernel_ms"]); k.append(s["kernel_ms"])
torch.cuda.synchronize()
print("%.3f ms/step, first pass %.3f ms, all kernels %.3f ms (%d steps, %dx%d, %d obs, mp %d)" % ((time.perf_counter() - t0) / steps * 1e3, np.mean(u), np.mean(k), steps, ny, ny, S, mp))
