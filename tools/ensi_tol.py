"""Largest relative deviation from the LAPACK golden vectors (tests/golden/ensi_cases.npz) over all cases, for the value of
GPP_ENSI_JTOL2 in the environment (the stopping threshold of the Jacobi sweeps): how much margin the threshold leaves to 1e-5."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import gridpp_amd as gridpp
import ensi_golden
worst = 0.0
for name, c in ensi_golden.CASES.items():
    h, v, w, mp, allow = c["params"]
    nb, ns = c["blat"].size, c["plat"].size
    c = dict(c)
    for k, n_ in (("belev", nb), ("blaf", nb), ("pelev", ns), ("plaf", ns)):
        c.setdefault(k, np.full(n_, np.nan, np.float32))
    points = gridpp.Points(c["plat"], c["plon"], c["pelev"], c["plaf"])
    E = c["background"].shape[1]
    if "expected" not in c or "params" not in c:
        continue
    if "shape" in c and c["shape"][0] > 0:
        Y, X = int(c["shape"][0]), int(c["shape"][1])
        grid = gridpp.Grid(c["blat"].reshape(Y, X), c["blon"].reshape(Y, X), c["belev"].reshape(Y, X), c["blaf"].reshape(Y, X))
        st = gridpp.BarnesStructure(grid, c["hfield"].reshape(Y, X), c["vfield"].reshape(Y, X), c["wfield"].reshape(Y, X)) if "hfield" in c else gridpp.BarnesStructure(h, v, w)
        out = gridpp.optimal_interpolation_ensi(grid, c["background"].reshape(Y, X, E), points, c["pobs"], c["psigmas"], c["pbackground"], st, int(mp), bool(allow))
    else:
        out = gridpp.optimal_interpolation_ensi(gridpp.Points(c["blat"], c["blon"], c["belev"], c["blaf"]), c["background"], points, c["pobs"], c["psigmas"],
                                                c["pbackground"], gridpp.BarnesStructure(h, v, w), int(mp), bool(allow))
    out = np.asarray(out); exp = c["expected"].reshape(out.shape)
    err = ensi_golden.rel_err(out, exp.astype(np.float64), c["background"]).max()
    worst = max(worst, err)
    print("%-32s %.2e" % (name, err))
print("JTOL2 %s worst %.2e" % (os.environ.get("GPP_ENSI_JTOL2", "default"), worst))
