"""Headline OI workload: per-call statistics (tiles left to the fallback kernel, factorisations, kernel ms)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gridpp_amd as gridpp
from bench import make_workload
ny = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
mp = int(sys.argv[3]) if len(sys.argv) > 3 else 30
lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(ny, ny, S, 1002, 0, ny)
grid = gridpp.Grid(lats, lons); points = gridpp.Points(plat, plon)
st = gridpp.BarnesStructure(10000)
dbg = bg
for i in range(3):
    out = gridpp.optimal_interpolation(grid, dbg, points, obs, ratios, pbg, st, mp)
    print(gridpp.oi_last_stats())
