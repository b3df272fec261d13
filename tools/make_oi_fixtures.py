#!/usr/bin/env python
"""Golden vectors for optimal_interpolation / optimal_interpolation_full (build container only; the output travels).

An INDEPENDENT restatement of /root/reference/src/api/oi.cpp:176-338 (the Points overload; a Grid is its row-major flattening,
oi.cpp:60-110) in numpy + scipy.linalg -- `inv` is LAPACK's dgetrf / dgetri, which is what `arma::inv` calls in the reference
build -- sharing no code with oracle/gridpp_oracle.c.  The reference's own OI tests hold 1-observation analytic cases, the
cross-validation identity and missing-value / extrapolation cases (tests/test_optimal_interpolation.py); the top-max_points cut,
the multi-observation inverse, the anti-extrapolation clamp on several observations and the analysis variance are pinned by
these vectors: tests/test_oracle_golden.py (CPU oracle) and tests/test_gpu_oi_parity.py (MI355X) check them to 1e-5.

Float semantics follow the C++ expression types line by line (float32 where the reference holds a `float`, float64 inside the
arma:: objects).  std::sort of (rho, index) pairs is not stable: the generator asserts that no grid point has two candidates with
the same rho at the cut, so that every implementation must select the same observations.

    python tools/make_oi_fixtures.py        ->  tests/golden/oi_cases.npz
"""
import os
import sys

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_ensi_fixtures import F, convert_coordinates, corr_barnes, get_neighbours, localization_distance   # noqa: E402


def cressman_rho(dist, length):
    """structure.cpp:35-44 on float32 arrays (signed `dist` for the elevation / laf factors: only dist >= length cuts)."""
    length = F(length)
    if not np.isfinite(length) or length == 0:
        return np.ones(dist.shape, F)
    l2 = F(length * length)
    d2 = (dist * dist).astype(F)
    r = ((l2 - d2).astype(F) / (l2 + d2).astype(F)).astype(F)
    r = np.where(dist >= length, F(0), r)
    return np.where(np.isfinite(dist), r, F(0)).astype(F)


def corr_cressman(p1, p2, h, v, w):
    """CressmanStructure::corr, structure.cpp:297-309; p1 one point, p2 arrays.  Not symmetric in (p1, p2) when v or w are set."""
    dx, dy, dz = (F(p1[0]) - p2[0]).astype(F), (F(p1[1]) - p2[1]).astype(F), (F(p1[2]) - p2[2]).astype(F)
    hdist = np.sqrt(((dx * dx).astype(F) + (dy * dy).astype(F)).astype(F) + (dz * dz).astype(F)).astype(F)
    rho = cressman_rho(hdist, h)
    if np.isfinite(p1[3]):
        rho = np.where(np.isfinite(p2[3]), (rho * cressman_rho((F(p1[3]) - p2[3]).astype(F), v)).astype(F), rho)
    if np.isfinite(p1[4]):
        rho = np.where(np.isfinite(p2[4]), (rho * cressman_rho((F(p1[4]) - p2[4]).astype(F), w)).astype(F), rho)
    return rho.astype(F), hdist


def oi_full(blat, blon, belev, blaf, background, bvariance, plat, plon, pelev, plaf, pobs, obs_variance, pbackground,
            bvariance_at_points, h, v, w, max_points, allow_extrapolation, kind="barnes"):
    background, bvariance = background.astype(F), bvariance.astype(F)
    pobs, obs_variance, pbackground, bvp = pobs.astype(F), obs_variance.astype(F), pbackground.astype(F), bvariance_at_points.astype(F)
    nY, nS = background.size, plat.size
    output, analysis_variance = background.copy(), bvariance.copy()                         # :197-199
    if nS == 0:
        return output, analysis_variance
    pratios = (obs_variance / bvp).astype(F)                                                   # :192-195
    bx, by, bz = convert_coordinates(blat, blon)
    px, py, pz = convert_coordinates(plat, plon)
    pelev, plaf = pelev.astype(F), plaf.astype(F)
    corr = corr_barnes if kind == "barnes" else corr_cressman
    loc = localization_distance(h) if kind == "barnes" else F(h)      # structure.cpp:280-282 / StructureFunction(h) for Cressman (:286)
    for y in range(nY):
        if not np.isfinite(background[y]):                                                     # :223
            continue
        p1 = (bx[y], by[y], bz[y], F(belev[y]), F(blaf[y]))
        idx0 = get_neighbours(px, py, pz, bx[y], by[y], bz[y], loc)                            # :233
        if idx0.size == 0:
            continue
        rhos, _ = corr(p1, (px[idx0], py[idx0], pz[idx0], pelev[idx0], plaf[idx0]), h, v, w)   # :250 (corr_background == corr here)
        keep = np.isfinite(pobs[idx0]) & np.isfinite(pbackground[idx0]) & (rhos > 0)           # :251-257
        cand, crho = idx0[keep], rhos[keep]
        if max_points > 0 and cand.size > max_points:                                          # :262-273: the largest rho first
            order = np.argsort(-crho.astype(np.float64), kind="stable")
            cut = crho[order]
            assert cut[max_points - 1] != cut[max_points], "rho tie at the max_points cut: the selection would be implementation defined"
            cand, crho = cand[order[:max_points]], crho[order[:max_points]]
        lS = cand.size
        if lS == 0:
            continue
        lObs, lY = pobs[cand].astype(np.float64), pbackground[cand].astype(np.float64)
        lG = crho.astype(np.float64)[None, :]
        lP = np.empty((lS, lS))
        for i in range(lS):                                                                    # :304-312 corr(p_i, p_j), float32 -> double
            k = cand[i]
            lP[i], _ = corr((px[k], py[k], pz[k], pelev[k], plaf[k]), (px[cand], py[cand], pz[cand], pelev[cand], plaf[cand]), h, v, w)
        lR = np.diag(pratios[cand].astype(np.float64))
        lGSR = lG @ sla.inv(lP + lR)                                                           # :315
        increment = F((lGSR @ (lObs - lY))[0])                                                 # :316-317
        if not allow_extrapolation:                                                            # :318-334
            maxInc, minInc = F((lObs - lY).max()), F((lObs - lY).min())
            if maxInc > 0 and increment > maxInc:
                increment = maxInc
            elif maxInc < 0 and increment > 0:
                increment = maxInc
            elif minInc < 0 and increment < minInc:
                increment = minInc
            elif minInc > 0 and increment < 0:
                increment = minInc
        output[y] = F(background[y] + increment)                                               # :335
        a00 = float((lGSR @ lG.T)[0, 0])
        analysis_variance[y] = F(np.float64(bvariance[y]) * (1.0 - a00))                        # :337
    return output, analysis_variance


def make_case(seed, Y, X, S, h, max_points, allow, v=0.0, w=0.0, elev=False, laf=False, nans=False, points_background=0, cluster=False,
              kind="barnes"):
    rng = np.random.default_rng(seed)
    if points_background:
        blat, blon = rng.random(points_background).astype(F), rng.random(points_background).astype(F)
        shape = np.array([0, points_background])
    else:
        lats, lons = np.meshgrid(np.linspace(0, 1, Y), np.linspace(0, 1, X), indexing="ij")
        blat, blon = lats.ravel().astype(F), lons.ravel().astype(F)
        shape = np.array([Y, X])
    n = blat.size
    bg = (np.sin(5 * blat) * np.cos(3 * blon) + rng.normal(0, 0.5, n)).astype(F)
    bvar = rng.uniform(0.5, 2.0, n).astype(F)
    belev = rng.uniform(0, 500, n).astype(F) if elev else np.full(n, np.nan, F)
    blaf = rng.uniform(0, 1, n).astype(F) if laf else np.full(n, np.nan, F)
    if cluster:      # a few dense blobs: strongly correlated observations, (P + R) far from diagonal
        cy, cx = rng.random(5), rng.random(5)
        k = rng.integers(0, 5, S)
        plat, plon = (cy[k] + 0.03 * rng.standard_normal(S)).astype(F), (cx[k] + 0.03 * rng.standard_normal(S)).astype(F)
    else:
        plat, plon = rng.random(S).astype(F), rng.random(S).astype(F)
    pelev = rng.uniform(0, 500, S).astype(F) if elev else np.full(S, np.nan, F)
    plaf = rng.uniform(0, 1, S).astype(F) if laf else np.full(S, np.nan, F)
    obs, pbg = rng.normal(0, 1, S).astype(F), rng.normal(0, 1, S).astype(F)
    ovar, bvp = rng.uniform(0.05, 2, S).astype(F), rng.uniform(0.5, 2, S).astype(F)
    if nans:
        obs[::9] = np.nan
        pbg[4::13] = np.nan
        bg[::17] = np.nan
    out, var = oi_full(blat, blon, belev, blaf, bg, bvar, plat, plon, pelev, plaf, obs, ovar, pbg, bvp, h, v, w, max_points, allow, kind)
    return dict(shape=shape, blat=blat, blon=blon, belev=belev, blaf=blaf, background=bg, bvariance=bvar, plat=plat, plon=plon,
                pelev=pelev, plaf=plaf, pobs=obs, obs_variance=ovar, pbackground=pbg, bvariance_at_points=bvp,
                params=np.array([h, v, w, max_points, 1.0 if allow else 0.0]), kind=np.array(0 if kind == "barnes" else 1),
                expected=out, expected_variance=var)


CASES = {
    "mp8_topk_cut":            dict(seed=31, Y=28, X=28, S=70, h=30000, max_points=8, allow=True),
    "mp8_noextrap":            dict(seed=32, Y=28, X=28, S=70, h=30000, max_points=8, allow=False),
    "mp12_elev_laf":           dict(seed=33, Y=24, X=24, S=80, h=25000, max_points=12, allow=True, v=300.0, w=0.5, elev=True, laf=True),
    "mp0_all_in_range":        dict(seed=34, Y=16, X=16, S=60, h=30000, max_points=0, allow=True),
    "mp30_clustered_noextrap": dict(seed=35, Y=20, X=20, S=200, h=30000, max_points=30, allow=False, cluster=True),
    "mp10_missing_values":     dict(seed=36, Y=24, X=24, S=90, h=30000, max_points=10, allow=False, nans=True),
    "mp6_points_background":   dict(seed=37, Y=0, X=0, S=50, h=25000, max_points=6, allow=True, points_background=400),
    "mp45_beyond_32":          dict(seed=38, Y=14, X=14, S=160, h=30000, max_points=45, allow=True),
    "mp5_short_range":         dict(seed=39, Y=28, X=28, S=60, h=5000, max_points=5, allow=True),     # many cells without observations
    # CressmanStructure: symmetric with h only; with v its factor acts on the SIGNED elevation difference, so (P + R) is not symmetric
    # and only a general inverse (LAPACK's LU here, the pivoted LU of k_oi on the GPU) gives the reference's numbers
    "cressman_mp10":           dict(seed=40, Y=24, X=24, S=90, h=60000, max_points=10, allow=True, kind="cressman"),
    "cressman_elev_mp12_noextrap": dict(seed=41, Y=20, X=20, S=90, h=60000, max_points=12, allow=False, v=400.0, elev=True, kind="cressman"),
}


def main():
    flat = {}
    for name, kw in CASES.items():
        d = make_case(**kw)
        for k, val in d.items():
            flat["%s/%s" % (name, k)] = val
        print("%-28s cells %5d  max |analysis - background| = %.3f  variance ratio min %.3f" % (
            name, d["blat"].size, np.nanmax(np.abs(d["expected"] - d["background"])), np.nanmin(d["expected_variance"] / d["bvariance"])))
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "oi_cases.npz")
    np.savez_compressed(dst, **flat)
    print("wrote", dst, os.path.getsize(dst) // 1024, "KB")


if __name__ == "__main__":
    sys.exit(main())
