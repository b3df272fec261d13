"""One headline OI call (or ny / obs / max_points given) and the statistics of it: factorisations, fallback tiles, kernel times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gridpp_amd as gridpp
from tools.bench_cases import make_workload
ny = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
mp = int(sys.argv[3]) if len(sys.argv) > 3 else 30
lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(ny, ny, S, 1002, 0, ny)
grid, points, st = gridpp.Grid(lats, lons), gridpp.Points(plat, plon), gridpp.BarnesStructure(10000)
d = [torch.from_numpy(a).cuda() for a in (bg, obs, ratios, pbg)]
for _ in range(3):
    out = gridpp.optimal_interpolation(grid, d[0], points, d[1], d[2], d[3], st, mp)
    print(gridpp.oi_last_stats())
