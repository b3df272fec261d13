"""Per-phase shader clocks of the EnSI kernels on config 5 (needs the profile build: tools/variant.sh ensiprof ensi -DGPP_ENSI_PROFILE -DGPP_TIMING_SWITCHES;
run with GPP_LIB=gridpp_amd/lib/var_ensiprof.so GPP_ENSI_STATS=1 [GPP_ENSI_MEMBERS2=1])."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gridpp_amd as gridpp
from bench_cases import ensi_inputs
lats, lons, bg, plat, plon, pbg, obs, sig = ensi_inputs(2500, 2500, 50, 5000)
grid, points, st = gridpp.Grid(lats, lons), gridpp.Points(plat, plon), gridpp.BarnesStructure(10000)
for k in range(2):
    r = gridpp.optimal_interpolation_ensi(grid, bg, points, obs, sig, pbg, st, 30)
    print("kernels %.1f ms" % gridpp.ensi_last_kernel_ms(), flush=True)
