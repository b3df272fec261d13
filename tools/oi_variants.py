"""The headline geometry through the other OI code paths (one line per variant)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gridpp_amd as gridpp
from bench import make_workload
ny = nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
S = 2500 * (ny // 2000) ** 2 if ny >= 2000 else 2500
lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(ny, nx, S, 1002, 0, ny)
grid, points = gridpp.Grid(lats, lons), gridpp.Points(plat, plon)
d = [torch.from_numpy(a).cuda() for a in (bg, obs, ratios, pbg)]
h = np.full((ny, nx), 10000.0, np.float32); z = np.zeros((ny, nx), np.float32)
variants = [
    ("Barnes(10000), max_points 30 (k_oi_union<true>)", gridpp.BarnesStructure(10000), 30),
    ("Cressman(10000), max_points 30 (k_oi_union<false>)", gridpp.CressmanStructure(10000), 30),
    ("Multiple(Barnes, Barnes, Barnes), max_points 30 (k_oi_union<false>)", gridpp.MultipleStructure(gridpp.BarnesStructure(10000), gridpp.BarnesStructure(10000, 100), gridpp.BarnesStructure(10000, 0, 0.5)), 30),
    ("CrossValidation(Barnes, 2000), max_points 30", gridpp.CrossValidation(gridpp.BarnesStructure(10000), 2000), 30),
    ("Barnes(10000), max_points 50 (k_oi<62>)", gridpp.BarnesStructure(10000), 50),
    ("Barnes(grid, h * ones, 0, 0): the 'var len scale' row of the reference's benchmark (tests/benchmark.py:66,294), max_points 30", gridpp.BarnesStructure(grid, h, z, z), 30),
    ("Barnes(grid, h, 0, 0) smoothly varying h (+-20 % over ~50 km), max_points 30 (k_oi_union_sp)", gridpp.BarnesStructure(grid, (h * (1 + 0.2 * np.sin(12 * lats) * np.cos(9 * lons))).astype(np.float32), z, z), 30),
    ("Barnes(grid, h, 0, 0) white-noise h (+-20 % per cell), max_points 30 (k_oi_union_sp)", gridpp.BarnesStructure(grid, (h * np.random.default_rng(3).uniform(0.8, 1.2, h.shape)).astype(np.float32), z, z), 30),
]
for name, st, mp in variants:
    f = lambda: gridpp.optimal_interpolation(grid, d[0], points, d[1], d[2], d[3], st, mp)
    f(); f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = gridpp.oi_last_stats()
    print(json.dumps({"variant": name, "grid": "%dx%d, %d obs" % (ny, nx, S), "ms": round(dt * 1e3, 3), "Mcells/s": round(ny * nx / dt / 1e6, 1), "union_ms": round(s["union_kernel_ms"], 3), "declined_tiles": s["fallback_tiles"], "factorisations": s["solves"]}), flush=True)
