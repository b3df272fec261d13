"""max_points beyond the 62-row register tile: timing of the large-n kernel (one workgroup per grid point)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gridpp_amd as gridpp
from bench import make_workload
for ny, S, mp in ((200, 2000, 100), (200, 2000, 200), (400, 3000, 100)):
    lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(ny, ny, S, 7, 0, ny)
    grid, points, st = gridpp.Grid(lats, lons), gridpp.Points(plat, plon), gridpp.BarnesStructure(10000)
    gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, mp)
    t0 = time.perf_counter(); out = gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, mp); dt = time.perf_counter() - t0
    s = gridpp.oi_last_stats()
    print("%dx%d grid, %d obs, max_points=%d: %.1f ms (%d cells on k_oi_big, %.1f us per cell)" % (ny, ny, S, mp, dt * 1e3, s["big_cells"], dt * 1e6 / max(s["big_cells"], 1)))
