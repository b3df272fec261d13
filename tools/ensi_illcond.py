"""EnSI off the benign manifold (round-4 verdict, item 1): a conditioning table for the tile path (max_points <= 32: k_ensi_pair + k_ensi_members,
default mode = early-stopped sweeps + float32 perturbation series, and converged mode) and for the large-n kernels (k_ensi_big_ns: Newton-Schulz;
k_ensi_huge with more than 64 members), against the oracle in the PLAIN measure |out - ref| / max(|ref|, 1e-2).

    python tools/ensi_illcond.py [> profiles/r05_ensi_illcond.txt]

Axes: observation sigmas x {1, 0.1, 0.01, 1e-3}, member spread x {0.01, 1, 100}; plus observations 20 spreads away from the ensemble and two
near-duplicate members.  Per cell of the table: worst deviation and how many float32 outputs differ from the oracle's at all (a value one ulp of
its operands off shows up there long before it leaves 1e-5).

Third column: the REFERENCE'S OWN reproducibility on the same inputs -- the oracle (pivoted LU inverse + cyclic Jacobi, oracle/gridpp_oracle.c)
against the independent numpy / LAPACK restatement of tools/make_ensi_fixtures.py (dgetrf / dgetri + dsyevd: what Armadillo calls), i.e. two
faithful implementations of oi_ensi.cpp:379-437.  The reference inverts Pinv = Y^T R^-1 Y + (E-1) I explicitly and takes eig_sym of the inverse:
with spread / sigma >= 1e3 cond(Pinv) passes 1e8 and the two differ by 1e-4 ... 1e-2 (at 1e5 LAPACK's eigenvalues of the inverse come out
negative: NaN), so a 1e-5 comparison with either of them means nothing there.  A row PASSES when the kernels are within 1e-5 of the oracle, or --
where the reference itself is not reproducible to 1e-6 -- within three times the distance between its two implementations.
Exit code 1 if a row fails."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gridpp_amd as gridpp
from oracle import oracle as O
from tests.test_gpu_ensi_parity import case
from tools import make_ensi_fixtures as LAP


def variant(c, sig_scale=1.0, spread=1.0, offset=0.0, dup=False):
    lats, lons, bg, plat, plon, pbg, obs, sig = [np.array(x, copy=True) for x in c]
    sig = (sig * sig_scale).astype(np.float32)
    if spread != 1.0:
        m = bg.mean(axis=2, keepdims=True); bg = (m + spread * (bg - m)).astype(np.float32)
        pm = pbg.mean(axis=1, keepdims=True); pbg = (pm + spread * (pbg - pm)).astype(np.float32)
    if offset:
        obs = (obs + offset * np.where(np.arange(obs.size) % 2, -1, 1)).astype(np.float32)
    if dup:
        E = bg.shape[2]
        bg[:, :, E - 1] = bg[:, :, 0] * np.float32(1 + 1e-6); pbg[:, E - 1] = pbg[:, 0] * np.float32(1 + 1e-6)
    return lats, lons, bg, plat, plon, pbg, obs, sig


def run(c, h, mp):
    lats, lons, bg, plat, plon, pbg, obs, sig = c
    Y, X, E = bg.shape
    ref = O.oi_ensi(O.Pts(lats.ravel(), lons.ravel()), bg.reshape(-1, E), O.Pts(plat, plon), obs, sig, pbg, O.Barnes(h), mp, True).reshape(Y, X, E)
    res = {}
    nanb, nanp = np.full(Y * X, np.nan, np.float32), np.full(plat.size, np.nan, np.float32)
    with np.errstate(all="ignore"):
        lap = LAP.ensi(lats.ravel().astype(np.float32), lons.ravel().astype(np.float32), nanb, nanb, bg.reshape(-1, E), plat.astype(np.float32), plon.astype(np.float32),
                       nanp, nanp, obs, sig, pbg, h, 0.0, 0.0, mp, True).reshape(Y, X, E)
    both = ~np.isnan(ref) & ~np.isnan(lap)
    res["reference"] = (float((np.abs(ref[both].astype(np.float64) - lap[both]) / np.maximum(np.abs(lap[both]), 1e-2)).max()) if both.any() else float("nan"),
                        int((ref[both] != lap[both]).sum()), int(both.sum()), int((np.isnan(lap) & ~np.isnan(ref)).sum()))
    for mode in ("default", "converged"):
        gridpp.ensi_set_convergence(mode == "converged")
        try:
            out = np.asarray(gridpp.optimal_interpolation_ensi(gridpp.Grid(lats, lons), bg, gridpp.Points(plat, plon), obs, sig, pbg, gridpp.BarnesStructure(h), mp, True))
        finally:
            gridpp.ensi_set_convergence(False)
        assert (np.isnan(out) == np.isnan(ref)).all()
        m = ~np.isnan(ref)
        e = np.abs(out[m].astype(np.float64) - ref[m]) / np.maximum(np.abs(ref[m]), 1e-2)
        res[mode] = (float(e.max()), int((out[m] != ref[m]).sum()), int(m.sum()))
    return res


paths = [("tile path (max_points 30, E = 50: k_ensi_pair + k_ensi_members)", 50, 30, 30000.0, 9, 9, 150),
         ("tile path (max_points 10, E = 20)", 20, 10, 30000.0, 9, 9, 150),
         ("large n, <= 48 members (max_points 50, E = 30: k_ensi_big_ns, tiles on and above the diagonal until round 5)", 30, 50, 60000.0, 7, 7, 150),
         ("large n, 49..64 members (max_points 0, E = 60: k_ensi_big_ns, strips)", 60, 0, 60000.0, 6, 6, 120),
         ("large n, more than 64 members (max_points 40, E = 70: k_ensi_huge)", 70, 40, 60000.0, 5, 5, 120)]
worst_all, failed = 0.0, []
for title, E, mp, h, Y, X, S in paths:
    base = case(31337 + E, Y, X, E, S)
    print("== %s, %d x %d grid, %d observations" % (title, Y, X, S))
    print("%-41s | %-36s | %-36s | %s" % ("input", "default mode: worst, floats differing", "converged mode: worst, floats differ.", "reference: oracle vs LAPACK restatement"))
    rows = [("sigma x %g, spread x %g" % (s_, p_), dict(sig_scale=s_, spread=p_)) for s_ in (1.0, 0.1, 0.01, 1e-3) for p_ in (0.01, 1.0, 100.0)]
    rows += [("observations +-20 spreads away", dict(offset=20.0)), ("two near-duplicate members", dict(dup=True)),
             ("sigma x 0.01, obs +-20, duplicate members", dict(sig_scale=0.01, offset=20.0, dup=True))]
    for name, kw in rows:
        r = run(variant(base, **kw), h, mp)
        gpu = max(r["default"][0], r["converged"][0])
        noise = r["reference"][0] if r["reference"][3] == 0 else float("inf")     # (LAPACK returned NaN where the oracle did not: no reference at all)
        ok = gpu < 1e-5 or (noise > 1e-6 and gpu <= 3 * noise)
        if noise <= 1e-6:
            worst_all = max(worst_all, gpu)
        if not ok:
            failed.append((title, name, gpu, noise))
        rtxt = "%.3g, %d of %d" % r["reference"][:3] + (" (+ %d NaN from LAPACK)" % r["reference"][3] if r["reference"][3] else "")
        print("%-41s | %-36s | %-36s | %s%s" % (name, "%.3g, %d of %d" % r["default"], "%.3g, %d of %d" % r["converged"], rtxt, "" if ok else "   <-- FAIL"), flush=True)
print("worst plain deviation from the oracle over the rows where the reference reproduces itself to 1e-6: %.3g; rows failing: %d -> %s" % (worst_all, len(failed), "PASS" if not failed else "FAIL"))
for f in failed:
    print("FAILED", f)
sys.exit(0 if not failed else 1)
