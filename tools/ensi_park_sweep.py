"""Config 5 with the park of k_ensi_pair -> k_ensi_members3 bounded to a few sizes (GPP_ENSI_PARK_MB): does the kernels' time depend on how much
memory the hand-over streams through (TLB reach, last-level cache) or only on the work?   python tools/ensi_park_sweep.py [MB ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gridpp_amd as gridpp
from bench_cases import ensi_inputs
lats, lons, bg, plat, plon, pbg, obs, sig = ensi_inputs(2500, 2500, 50, 5000)
grid, points, st = gridpp.Grid(lats, lons), gridpp.Points(plat, plon), gridpp.BarnesStructure(10000)
sizes = [int(x) for x in sys.argv[1:]] or [256, 1024, 4096, 16384, 0]
for mb in sizes:
    gridpp.set_path_override("GPP_ENSI_PARK_MB", str(mb) if mb else None)
    gridpp.release_workspaces()
    ms = []
    for k in range(3):
        gridpp.optimal_interpolation_ensi(grid, bg, points, obs, sig, pbg, st, 30)
        ms.append(gridpp.ensi_last_kernel_ms())
    print("park %6s MB: kernels %s ms" % (mb if mb else "default", " ".join("%.1f" % m for m in ms)), flush=True)
gridpp.set_path_override("GPP_ENSI_PARK_MB", None)
