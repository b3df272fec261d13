"""Cost of the observation index of a Points object (built on the device at its first use in an optimal_interpolation call, kept with the
object): first call against second call on the headline geometry's 500-row slice -- what every rank of an N-GPU run pays once."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gridpp_amd as gridpp
from bench import make_workload
lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(4000, 4000, 10000, 1002, 0, 500)
grid, st = gridpp.Grid(lats, lons), gridpp.BarnesStructure(10000)
d = [torch.from_numpy(a).cuda() for a in (bg, obs, ratios, pbg)]
warm = gridpp.Points(plat, plon)
for _ in range(3): gridpp.optimal_interpolation(grid, d[0], warm, d[1], d[2], d[3], st, 30)
ts = []
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pts = gridpp.Points(plat, plon)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    gridpp.optimal_interpolation(grid, d[0], pts, d[1], d[2], d[3], st, 30)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    gridpp.optimal_interpolation(grid, d[0], pts, d[1], d[2], d[3], st, 30)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
best = [min(t[i] for t in ts) for i in range(3)]
print("Points(10000 obs): %.3f ms; first call with it: %.3f ms; second call: %.3f ms -> index build + first-use costs %.3f ms once per Points object"
      % (best[0], best[1], best[2], best[1] - best[2]))
