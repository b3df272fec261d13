"""The 2000 x 2000 / 2 500 observations geometry of tools/oi_variants.py for a list of max_points: ms per call and the first pass's (which kernel takes which max_points: DESIGN 4.1)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gridpp_amd as gridpp
from bench import make_workload
ny = nx = 2000
lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(ny, nx, 2500, 1002, 0, ny)
grid, points, st = gridpp.Grid(lats, lons), gridpp.Points(plat, plon), gridpp.BarnesStructure(10000)
d = [torch.from_numpy(a).cuda() for a in (bg, obs, ratios, pbg)]
for mp in [int(x) for x in (sys.argv[1:] or ["30", "33", "36", "40", "44", "46", "47", "50", "62"])]:
    f = lambda: gridpp.optimal_interpolation(grid, d[0], points, d[1], d[2], d[3], st, mp)
    f(); f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(); f(); f(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    s = gridpp.oi_last_stats()
    print("max_points %2d: %.3f ms per call, first pass %.3f ms, declined tiles %d, factorisations %d" % (mp, dt * 1e3, s["union_kernel_ms"], s["fallback_tiles"], s["solves"]), flush=True)
