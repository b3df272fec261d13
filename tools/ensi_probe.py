"""Four C5 EnSI calls with the wall time of each (first-call allocations, the kept park): for timing experiments and rocprofv3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gridpp_amd as gridpp
from tools.bench_cases import ensi_inputs
lats, lons, bg, plat, plon, pbg, obs, sig = ensi_inputs(2500, 2500, 50, 5000)
grid, points, st = gridpp.Grid(lats, lons), gridpp.Points(plat, plon), gridpp.BarnesStructure(10000)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gridpp.optimal_interpolation_ensi(grid, bg, points, obs, sig, pbg, st, 30)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("call %d: wall %.1f ms, kernel events %.1f ms, free %.1f GB" % (i, dt * 1e3, gridpp.ensi_last_kernel_ms(), torch.cuda.mem_get_info()[0] / 2**30))
