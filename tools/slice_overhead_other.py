"""One rank's share of configs 5 (EnSI 2500 x 2500 x 50, 5 000 obs, max_points 30) and 4 (neighbourhood Mean / quantile_fast, 4000 x 4000 x 100, halfwidth 15)
on an N-GPU run, measured on one GPU: the row tile of 2500 / N resp. 4000 / N rows (the neighbourhood tile with its `halfwidth`-row halos: the
interior ranks' case), wall time per step -> a bound on the strong-scaling speed-up before the RCCL broadcast / halo exchange (tools/slice_overhead.py
does the same for the headline)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gridpp_amd as gridpp
from tools.bench_cases import ensi_inputs


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


base = {}
only = sys.argv[1] if len(sys.argv) > 1 else ""
for N in (() if only == "nbh" else (1, 2, 4, 8)):
    rows = 2500 // N
    lats, lons, bg, plat, plon, pbg, obs, sig = ensi_inputs(2500, 2500, 50, 5000, 0, rows)
    grid, points, st = gridpp.Grid(lats, lons), gridpp.Points(plat, plon), gridpp.BarnesStructure(10000)
    t = timeit(lambda: gridpp.optimal_interpolation_ensi(grid, bg, points, obs, sig, pbg, st, 30), reps=2)
    base.setdefault("ensi", t)
    print("C5 EnSI  N=%d rows=%4d: %8.2f ms/step (kernels %.2f ms), speed-up bound %.2f" % (N, rows, t, gridpp.ensi_last_kernel_ms(), base["ensi"] / t), flush=True)
    del bg
    gridpp.release_workspaces() if hasattr(gridpp, "release_workspaces") else None
thr = torch.linspace(0, 10, 11, device="cuda")
for N in (() if only == "ensi" else (1, 2, 4, 8)):
    rows = 4000 // N + (30 if N > 1 else 0)          # an interior rank: its tile + a halo of `halfwidth` rows above and below
    g = torch.Generator(device="cuda").manual_seed(1003)
    cube = torch.rand((rows, 4000, 100), generator=g, device="cuda") * 10
    tm = timeit(lambda: gridpp.neighbourhood(cube, 15, gridpp.Mean))
    tq = timeit(lambda: gridpp.neighbourhood_quantile_fast(cube, 0.5, 15, thr))
    base.setdefault("mean", tm); base.setdefault("qf", tq)
    print("C4 nbh   N=%d rows=%4d (with halo): Mean %6.3f ms (bound %.2f), quantile_fast %6.3f ms (bound %.2f)" % (N, rows, tm, base["mean"] / tm, tq, base["qf"] / tq), flush=True)
    del cube
