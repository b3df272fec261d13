"""A/B of the two ensemble-side kernels of optimal_interpolation_ensi on config 5 (2500 x 2500 x 50 members, 5 000 observations, max_points 30):
k_ensi_members3 (three waves per SIMD, ensi_members3.h; the default) against k_ensi_members (two staging areas, two waves per SIMD;
GPP_ENSI_MEMBERS2).  Prints kernel-sum ms of both, in both convergence modes, and how far the two outputs are apart.
    python tools/ensi_members_ab.py [ny nx [E]]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gridpp_amd as gridpp
from bench_cases import ensi_inputs

ny = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
E = int(sys.argv[3]) if len(sys.argv) > 3 else 50
S, mp = 5000, 30
lats, lons, bg, plat, plon, pbg, obs, sig = ensi_inputs(ny, nx, E, S)
grid, points, st = gridpp.Grid(lats, lons), gridpp.Points(plat, plon), gridpp.BarnesStructure(10000)
out = {}
for conv in (False, True):
    gridpp.ensi_set_convergence(conv)
    for name, sw in (("members3", None), ("members2", "1")):
        gridpp.set_path_override("GPP_ENSI_MEMBERS2", sw)
        ms = []
        for rep in range(3):
            r = gridpp.optimal_interpolation_ensi(grid, bg, points, obs, sig, pbg, st, mp)
            ms.append(gridpp.ensi_last_kernel_ms())
        out[(conv, name)] = r.cpu().numpy() if hasattr(r, "cpu") else np.array(r)
        print("%-9s %-10s kernels %s ms" % ("converged" if conv else "default", name, " ".join("%.1f" % m for m in ms)), flush=True)
    a, b = out[(conv, "members3")], out[(conv, "members2")]
    d = np.abs(a - b) / np.maximum(np.abs(b), 1e-2)
    print("   members3 against members2: %d of %d values differ, worst %.3g relative (plain measure), NaN %d / %d" % (
        int((a != b).sum()), a.size, float(np.nanmax(d)), int(np.isnan(a).sum()), int(np.isnan(b).sum())), flush=True)
gridpp.set_path_override("GPP_ENSI_MEMBERS2", None)
gridpp.ensi_set_convergence(False)
