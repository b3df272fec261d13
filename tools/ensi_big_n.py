"""optimal_interpolation_ensi with max_points beyond the 32-row tile: every grid point with more than 32 usable observations
takes k_ensi_big (one workgroup per cell).  Prints wall time and the share of such cells."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gridpp_amd as gridpp
rng = np.random.default_rng(5)
for (Y, E, S, mp) in ((200, 50, 2000, 50), (500, 50, 5000, 50), (500, 30, 700, 0)):
    lats, lons = np.meshgrid(np.linspace(0, 1, Y), np.linspace(0, 1, Y), indexing="ij")
    grid = gridpp.Grid(lats, lons)
    pts = gridpp.Points(rng.random(S), rng.random(S))
    bg = torch.randn((Y, Y, E), device="cuda")
    obs, sig, pbg = torch.randn(S, device="cuda"), torch.rand(S, device="cuda") + 0.5, torch.randn((S, E), device="cuda")
    st = gridpp.BarnesStructure(10000)
    f = lambda: gridpp.optimal_interpolation_ensi(grid, bg, pts, obs, sig, pbg, st, mp)
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = f(); torch.cuda.synchronize(); t = time.perf_counter() - t0
    n32 = (gridpp.count(pts, grid, st.localization_distance()) > 32).mean()
    print("EnSI %dx%dx%d, %d obs, max_points=%d: %.1f ms (%.2f Mcells/s); cells with more than 32 observations in range: %.0f%%"
          % (Y, Y, E, S, mp, t * 1e3, Y * Y / t / 1e6, 100 * n32), flush=True)
