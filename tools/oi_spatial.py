"""The spatially varying BarnesStructure row of the reference's benchmark on a 2000 x 2000 grid (tests/benchmark.py:66 shape, scaled):
three calls, for rocprofv3 / timing."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gridpp_amd as gridpp
from bench import make_workload
ny = nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
S = 2500
lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(ny, nx, S, 1002, 0, ny)
grid, points = gridpp.Grid(lats, lons), gridpp.Points(plat, plon)
d = [torch.from_numpy(a).cuda() for a in (bg, obs, ratios, pbg)]
rng = np.random.default_rng(3)
h = np.full((ny, nx), 10000.0, np.float32) if len(sys.argv) < 3 else (10000 * rng.uniform(0.8, 1.2, (ny, nx))).astype(np.float32)
z = np.zeros((ny, nx), np.float32)
st = gridpp.BarnesStructure(grid, h, z, z)
f = lambda: gridpp.optimal_interpolation(grid, d[0], points, d[1], d[2], d[3], st, 30)
f(); torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter(); f(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"case": "spatially varying Barnes %dx%d, %d obs, max_points 30" % (ny, nx, S), "ms": round(dt * 1e3, 3), "stats": gridpp.oi_last_stats()}))
