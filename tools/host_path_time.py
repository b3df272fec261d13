"""Per-call time of the HOST path (numpy arrays in, numpy array out -- what a gridpp script does): optimal_interpolation on configs 1 and 2 and the headline,
neighbourhood(Mean, halfwidth 7) on 200^2 / 1000^2 / 4000^2 planes.  The staging buffers of such a call come from a pool since round 5 (csrc/common.h)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gridpp_amd as gridpp
from tools.bench_cases import make_workload


def per_call(fn, reps):
    for _ in range(3):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


for name, ny, S, mp, reps in (("C1 200^2, 10 obs, mp 10", 200, 10, 10, 200), ("C2 1000^2, 1k obs, mp 20", 1000, 1000, 20, 50), ("C3 4000^2, 10k obs, mp 30", 4000, 10000, 30, 10)):
    lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(ny, ny, S, 1002, 0, ny)
    grid, points, st = gridpp.Grid(lats, lons), gridpp.Points(plat, plon), gridpp.BarnesStructure(10000)
    bg64, obs64, ratios64, pbg64 = (a.astype(np.float64) for a in (bg, obs, ratios, pbg))     # (round 6: converted ONCE -- until then numpy's astype of 16 M values sat inside the timed call: 14 of the "21 ms")
    print("optimal_interpolation %-26s %8.3f ms per call (float32 arrays), %8.3f ms (float64 arrays)" % (
        name, per_call(lambda: gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, mp), reps),
        per_call(lambda: gridpp.optimal_interpolation(grid, bg64, points, obs64, ratios64, pbg64, st, mp), reps)), flush=True)
for n, reps in ((200, 300), (1000, 100), (4000, 10)):
    f = np.random.default_rng(1).uniform(0, 10, (n, n)).astype(np.float32)
    print("neighbourhood(Mean, halfwidth 7) %4d^2 %8.3f ms per call" % (n, per_call(lambda: gridpp.neighbourhood(f, 7, gridpp.Mean), reps)), flush=True)
