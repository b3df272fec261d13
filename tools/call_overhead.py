"""Fixed cost of one optimal_interpolation call on device-resident inputs (tiny problem: everything is overhead)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gridpp_amd as gridpp
lats, lons = np.meshgrid(np.linspace(0, 0.01, 8), np.linspace(0, 0.01, 8), indexing="ij")
grid = gridpp.Grid(lats, lons); pts = gridpp.Points(np.array([0.005]), np.array([0.005]))
st = gridpp.BarnesStructure(10000)
d = [torch.zeros((8, 8), device="cuda"), torch.zeros(1, device="cuda"), torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")]
f = lambda: gridpp.optimal_interpolation(grid, d[0], pts, d[1], d[2], d[3], st, 30)
for _ in range(50): f()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(500): f()
torch.cuda.synchronize(); print("per call: %.1f us (kernel events: %.1f us)" % ((time.perf_counter() - t0) / 500 * 1e6, gridpp.oi_last_stats()["kernel_ms"] * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(300): f()
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(8)
