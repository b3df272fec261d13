#!/usr/bin/env python
"""Golden vectors for optimal_interpolation_ensi (build container only; the output travels, this script's inputs do not).

An INDEPENDENT restatement of /root/reference/src/api/oi_ensi.cpp:114-568 in numpy + scipy.linalg -- i.e. with the LAPACK
routines (`dgetrf/dgetri` behind `inv`, `dsyev*` behind `eigh`, `dgecon` behind rcond) that Armadillo calls in the reference
build -- sharing no code with oracle/gridpp_oracle.c (whose EnSI part uses its own LU inverse and cyclic Jacobi).  The
reference's tests hold no numeric EnSI value (tests/test_optimal_interpolation_ens.py:9-35), and the reference cannot be built
here (no Boost / Armadillo), so these vectors are what pins the oracle's EnSI part: tests/test_oracle_golden.py (CPU) and
tests/test_gpu_ensi_parity.py (MI355X) check them to 1e-5.

Float semantics follow the C++ expression types line by line (float32 where the reference holds a `float`, float64 inside
the arma:: objects).  Every case is built so that the implementation-defined parts of the reference do not matter:
  * candidate order (R-tree traversal order) only matters through the `lY[e]` linear index of the anti-extrapolation
    filter (oi_ensi.cpp:523-524) when the reference did NOT sort, so the no-extrapolation cases either have more usable
    observations than max_points at every grid point (then the order is "rho descending") -- asserted below -- or, where
    that is the point of the case (max_points = 0), use index order, which is what oracle and kernels define;
  * std::sort ties: the generator asserts that no two candidates of a grid point share a rho at the cut.

    python tools/make_ensi_fixtures.py        ->  tests/golden/ensi_cases.npz
"""
import os
import sys

import numpy as np
import scipy.linalg as sla

F = np.float32
RADIUS_EARTH = 6.378137e6          # include/gridpp.h:55 (double)
DEFAULT_MIN_RHO = F(0.0013)        # src/api/structure.cpp (StructureFunction::default_min_rho)


def convert_coordinates(lats, lons):
    """util.cpp:606-612 (Geodetic): double trig, stored as float."""
    lonr = np.pi / 180 * lons.astype(F).astype(np.float64)
    latr = np.pi / 180 * lats.astype(F).astype(np.float64)
    return ((np.cos(latr) * np.cos(lonr) * RADIUS_EARTH).astype(F), (np.cos(latr) * np.sin(lonr) * RADIUS_EARTH).astype(F),
            (np.sin(latr) * RADIUS_EARTH).astype(F))


def barnes_rho(dist, length):
    """structure.cpp:26-34 on float32 arrays: v float, exponent and exp in double, result float."""
    length = F(length)
    if not np.isfinite(length) or length == 0:
        return np.ones(dist.shape, F)
    v = (dist / length).astype(F)
    r = np.exp(-0.5 * v.astype(np.float64) * v.astype(np.float64)).astype(F)
    return np.where(np.isfinite(dist), r, F(0)).astype(F)


def _loc_readings(h, min_rho):
    a = F(np.sqrt(np.float64(-2.0) * np.log(np.float64(min_rho))) * np.float64(F(h)))
    b = F(F(np.sqrt(F(F(-2) * np.log(F(min_rho))))) * F(h))
    return a, b


def _loc_readings_agree(h, min_rho=None):
    a, b = _loc_readings(h, DEFAULT_MIN_RHO if min_rho is None else min_rho)
    return a == b


def localization_distance(h, min_rho=DEFAULT_MIN_RHO):
    """structure.cpp:280-282.  The float / double reading of log and sqrt must not matter for a fixture."""
    a, b = _loc_readings(h, min_rho)
    assert a == b, (a, b)
    return a


def corr_barnes(p1, p2, h, v, w):
    """BarnesStructure::corr (scalar form), structure.cpp:215-228; p1 one point, p2 arrays.  Points are (x, y, z, elev, laf)."""
    dx, dy, dz = (F(p1[0]) - p2[0]).astype(F), (F(p1[1]) - p2[1]).astype(F), (F(p1[2]) - p2[2]).astype(F)
    hdist = np.sqrt(((dx * dx).astype(F) + (dy * dy).astype(F)).astype(F) + (dz * dz).astype(F)).astype(F)   # kdtree.cpp:192-194
    rho = barnes_rho(hdist, h)
    if np.isfinite(p1[3]):
        ok = np.isfinite(p2[3])
        rho = np.where(ok, (rho * barnes_rho((F(p1[3]) - p2[3]).astype(F), v)).astype(F), rho)
    if np.isfinite(p1[4]):
        ok = np.isfinite(p2[4])
        rho = np.where(ok, (rho * barnes_rho((F(p1[4]) - p2[4]).astype(F), w)).astype(F), rho)
    return np.where(hdist > localization_distance(h), F(0), rho).astype(F), hdist


def get_neighbours(px, py, pz, x, y, z, radius):
    """kdtree.cpp:39-60,241-260: strictly inside the box, chord <= radius; returned in index order."""
    r = F(radius)
    inside = (px > F(x - r)) & (px < F(x + r)) & (py > F(y - r)) & (py < F(y + r)) & (pz > F(z - r)) & (pz < F(z + r))
    dx, dy, dz = (px - F(x)).astype(F), (py - F(y)).astype(F), (pz - F(z)).astype(F)
    d = np.sqrt(((dx * dx).astype(F) + (dy * dy).astype(F)).astype(F) + (dz * dz).astype(F)).astype(F)
    return np.nonzero(inside & (d <= r))[0]


def seq_mean(row):
    """calc_statistic(Mean), util.cpp:22-38: sequential float sum over the valid values."""
    ok = np.isfinite(row)
    if not ok.any():
        return F(np.nan)
    return F(np.cumsum(row[ok].astype(F), dtype=F)[-1] / F(ok.sum()))


def ensi(blat, blon, belev, blaf, background, plat, plon, pelev, plaf, pobs, psigmas, pbackground, h, v, w, max_points,
         allow_extrapolation, info=None):
    """oi_ensi.cpp:114-568 (Points overload; a Grid is its row-major flattening, :69-110)."""
    background = background.astype(F)
    nY, nEns = background.shape
    nS = plat.size
    out = background.copy()
    if nS == 0:
        return out
    bx, by, bz = convert_coordinates(blat, blon)
    px, py, pz = convert_coordinates(plat, plon)
    pobs, psigmas, pbackground = pobs.astype(F), psigmas.astype(F), pbackground.astype(F)
    # :166-178
    gY = pbackground.copy()
    gYhat = np.empty(nS, F)
    for i in range(nS):
        m = seq_mean(gY[i])
        if np.isfinite(m):
            ok = np.isfinite(gY[i])
            gY[i, ok] = (gY[i, ok] - m).astype(F)
        gYhat[i] = m
    # :187-201
    validEns = [e for e in range(nEns) if np.isfinite(background[:, e]).all()]
    nV = len(validEns)
    # h, v, w: scalars, or one value per background point (the spatially varying BarnesStructure(grid, h, v, w) on the background's
    # own grid, structure.cpp:168-214: localization_distance(p1) and corr_background(p1, .) take the scales at p1)
    hs, vs, ws = (np.broadcast_to(np.asarray(t, F), (nY,)) for t in (h, v, w))
    sorted_everywhere = True
    for y in range(nY):
        h, v, w = float(hs[y]), float(vs[y]), float(ws[y])
        loc = localization_distance(h)
        p1 = (bx[y], by[y], bz[y], F(belev[y]), F(blaf[y]))
        idx0 = get_neighbours(px, py, pz, bx[y], by[y], bz[y], loc)                       # :213
        if idx0.size == 0:
            continue
        rhos, _ = corr_barnes(p1, (px[idx0], py[idx0], pz[idx0], pelev[idx0].astype(F), plaf[idx0].astype(F)), h, v, w)   # :227
        keep = np.isfinite(pobs[idx0]) & (rhos > 0)                                      # :230-237
        cand, crho = idx0[keep], rhos[keep]
        if max_points > 0 and cand.size > max_points:                                    # :241-255
            order = np.argsort(crho, kind="stable")[::-1]
            assert crho[order[max_points - 1]] != crho[order[max_points]], "rho tie at the cut: choose another seed"
            assert np.unique(crho[order[:max_points]]).size == max_points, "rho tie inside the selection"
            sel, lRhos = cand[order[:max_points]], crho[order[:max_points]].astype(np.float64)
        else:
            sel, lRhos = cand, crho.astype(np.float64)
            sorted_everywhere = False
        lS = sel.size
        if lS == 0:
            continue
        lObs = pobs[sel].astype(np.float64)
        lY = gY[sel][:, validEns].astype(np.float64)                                      # :282-296  (lS x nV)
        lYhat = gYhat[sel].astype(np.float64)
        Rinv = np.diag(lRhos / (psigmas[sel] * psigmas[sel]).astype(F).astype(np.float64))   # :300
        C = lY.T @ Rinv                                                                   # :380
        diag = F(F(1) / F(1) * F(nV - 1))                                                 # :383
        Pinv = C @ lY + np.float64(diag) * np.eye(nV)                                     # :385
        if nV == 0:
            continue
        # :386 arma::rcond -> LAPACK dgecon on the LU factors (1-norm)
        lu, piv, _ = sla.lapack.dgetrf(Pinv)
        rcond = sla.lapack.dgecon(lu, np.linalg.norm(Pinv, 1))[0]
        if F(rcond) <= 0:
            continue
        P = sla.inv(Pinv)                                                                 # :399
        eigval, eigvec = sla.eigh(np.float64(nV - 1) * P)                                 # :402
        W = eigvec @ np.diag(np.sqrt(eigval)) @ eigvec.T                                  # :423-425
        PC = P @ C                                                                        # :433-434
        wv = PC @ (lObs - lYhat)                                                          # :437-441
        W = W + wv[:, None]                                                               # :444-448  W(e, e2) += w(e)
        vals = background[y, validEns]
        ensMean = F(np.cumsum(vals, dtype=F)[-1] / F(nV))                                 # :451-462
        X = vals.astype(np.float64) - np.float64(ensMean)                                 # :463-465
        total = np.zeros(nV, F)
        for k in range(nV):                                                               # :505-511 float accumulation
            total = (total.astype(np.float64) + X[k] * W[k, :]).astype(F)
        curr = total.copy()
        if not allow_extrapolation:                                                       # :520-552
            lYflat = lY.flatten(order="F")                                                # lY[e]: LINEAR column-major index
            for e in range(nV):
                incs = lObs - (lYflat[e] + lYhat)
                maxInc, minInc = F(incs.max()), F(incs.min())
                member = F(np.float64(curr[e]) - X[e])
                if maxInc > 0 and member > maxInc:
                    curr[e] = F(np.float64(maxInc) + X[e])
                elif maxInc < 0 and member > 0:
                    curr[e] = F(0 + X[e])
                elif minInc < 0 and member < minInc:
                    curr[e] = F(np.float64(minInc) + X[e])
                elif minInc > 0 and member < 0:
                    curr[e] = F(0 + X[e])
        out[y, validEns] = (ensMean + curr).astype(F)                                     # :553
    if info is not None:
        info["sorted_everywhere"] = sorted_everywhere
    return out


def one_obs_closed_form(bg_row, y_row, rho, sigma, obs, yhat):
    """Analytic 1-observation EnSI update (all members valid), in double: with Y = y^T (1 x E), r = rho / sigma^2, c = E - 1:
    Pinv = r y y^T + c I  =>  P = (I - r y y^T / (c + r |y|^2)) / c,   sqrt(c P) = I - beta y y^T / |y|^2 with
    beta = 1 - sqrt(c / (c + r |y|^2)),   w = r (obs - yhat) y / (c + r |y|^2).   out = mean + X (sqrt(cP) + w 1^T)."""
    E = bg_row.size
    c = float(E - 1)
    y = y_row.astype(np.float64)
    r = float(rho) / float(sigma) ** 2
    yy = float(y @ y)
    beta = 1.0 - np.sqrt(c / (c + r * yy))
    wv = r * (float(obs) - float(yhat)) * y / (c + r * yy)
    mean = bg_row.astype(np.float64).mean()
    X = bg_row.astype(np.float64) - mean
    return mean + X - beta * (X @ y) * y / yy + (X @ wv)


def make_case(seed, Y, X, E, S, h, max_points, allow, v=0.0, w=0.0, elev=False, laf=False, nan_member=None, nan_obs=False,
              points_background=0, spatial=False):
    rng = np.random.default_rng(seed)
    if points_background:
        blat, blon = rng.random(points_background).astype(F), rng.random(points_background).astype(F)
        base = np.sin(5 * blat) * np.cos(3 * blon)
        bg = (base[:, None] + rng.normal(0, 1, (points_background, E))).astype(F)
    else:
        lats, lons = np.meshgrid(np.linspace(0, 1, Y), np.linspace(0, 1, X), indexing="ij")
        blat, blon = lats.ravel().astype(F), lons.ravel().astype(F)
        base = np.sin(5 * blat) * np.cos(3 * blon)
        bg = (base[:, None] + rng.normal(0, 1, (Y * X, E))).astype(F)
    n = blat.size
    belev = rng.uniform(0, 500, n).astype(F) if elev else np.full(n, np.nan, F)
    blaf = rng.uniform(0, 1, n).astype(F) if laf else np.full(n, np.nan, F)
    plat, plon = rng.random(S).astype(F), rng.random(S).astype(F)
    pelev = rng.uniform(0, 500, S).astype(F) if elev else np.full(S, np.nan, F)
    plaf = rng.uniform(0, 1, S).astype(F) if laf else np.full(S, np.nan, F)
    pbg = rng.normal(0, 1, (S, E)).astype(F)
    obs = rng.normal(0, 1, S).astype(F)
    sig = rng.uniform(0.5, 2, S).astype(F)
    if nan_member is not None:
        bg[n // 3, nan_member] = np.nan
    if nan_obs:
        obs[::7] = np.nan
    info = {}
    extra = {}
    if spatial:     # per-cell scales (+- 30 % around the nominal ones)
        hq = (np.round(h * rng.uniform(0.7, 1.3, n) / 250.0) * 250.0).astype(F)
        for i in range(n):      # keep only scales whose localisation distance reads the same in float and in double
            while not _loc_readings_agree(hq[i]):
                hq[i] += F(250.0)
        vq = (v * rng.uniform(0.7, 1.3, n)).astype(F)
        wq = (w * rng.uniform(0.7, 1.3, n)).astype(F)
        extra = dict(hfield=hq, vfield=vq, wfield=wq)
        out = ensi(blat, blon, belev, blaf, bg, plat, plon, pelev, plaf, obs, sig, pbg, hq, vq, wq, max_points, allow, info)
    else:
        out = ensi(blat, blon, belev, blaf, bg, plat, plon, pelev, plaf, obs, sig, pbg, h, v, w, max_points, allow, info)
    if not allow and max_points > 0:
        assert info["sorted_everywhere"], "no-extrapolation case with an unsorted grid point: its result is order dependent"
    d = dict(shape=np.array([Y, X, E] if not points_background else [0, points_background, E]), blat=blat, blon=blon, belev=belev,
             blaf=blaf, background=bg, plat=plat, plon=plon, pelev=pelev, plaf=plaf, pobs=obs, psigmas=sig, pbackground=pbg,
             params=np.array([h, v, w, max_points, 1.0 if allow else 0.0]), expected=out)
    d.update(extra)
    return d


CASES = {
    # name: kwargs of make_case
    "e3_mp5":            dict(seed=11, Y=32, X=32, E=3, S=40, h=30000, max_points=5, allow=True),
    "e10_mp10_noextrap": dict(seed=12, Y=32, X=32, E=10, S=60, h=30000, max_points=10, allow=False),
    "e50_mp30":          dict(seed=13, Y=20, X=24, E=50, S=80, h=30000, max_points=30, allow=True),
    "e50_mp30_noextrap": dict(seed=14, Y=12, X=12, E=50, S=80, h=35000, max_points=30, allow=False),
    "e10_nanmember_nanobs_noextrap": dict(seed=15, Y=32, X=32, E=10, S=90, h=35000, max_points=12, allow=False, nan_member=2, nan_obs=True),
    "e10_elev_laf":      dict(seed=16, Y=32, X=32, E=10, S=60, h=25000, max_points=8, allow=True, v=300.0, w=0.5, elev=True, laf=True),
    "e12_mp0_all_obs":   dict(seed=17, Y=16, X=16, E=12, S=120, h=30000, max_points=0, allow=True),
    "e20_mp45":          dict(seed=18, Y=16, X=16, E=20, S=150, h=30000, max_points=45, allow=True),
    "e80_mp20":          dict(seed=19, Y=12, X=12, E=80, S=60, h=30000, max_points=20, allow=True),
    "e6_points_background": dict(seed=20, Y=0, X=0, E=6, S=40, h=25000, max_points=8, allow=True, points_background=300),
    "e8_short_range":    dict(seed=21, Y=32, X=32, E=8, S=50, h=5000, max_points=6, allow=True),   # many cells without observations
    # spatially varying BarnesStructure(grid, h, v, w) on the background grid (GPU tests only: the oracle's EnSI part takes scalar scales)
    "e10_spatial_mp10":  dict(seed=22, Y=24, X=24, E=10, S=70, h=25000, max_points=10, allow=True, v=300.0, w=0.5, elev=True, laf=True, spatial=True),
    "e12_spatial_mp0":   dict(seed=23, Y=14, X=14, E=12, S=110, h=28000, max_points=0, allow=True, v=300.0, elev=True, spatial=True),
}


def main():
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ensi_cases.npz")
    flat = {}
    for name, kw in CASES.items():
        d = make_case(**kw)
        changed = np.nanmax(np.abs(d["expected"] - d["background"]))
        print("%-34s cells %5d  E %3d  max |analysis - background| = %.3f" % (name, d["blat"].size, d["background"].shape[1], changed), flush=True)
        for k, val in d.items():
            flat[name + "/" + k] = val
    # analytic 1-observation case: checked against the restatement here, stored with its closed-form answer
    rng = np.random.default_rng(30)
    E = 7
    blat, blon = np.array([0.50, 0.52, 0.47, 0.5], F), np.array([0.50, 0.49, 0.53, 0.95], F)
    bg = rng.normal(2, 1, (4, E)).astype(F)
    plat, plon = np.array([0.505], F), np.array([0.495], F)
    pbg = rng.normal(1, 1, (1, E)).astype(F)
    obs, sig = np.array([2.5], F), np.array([0.8], F)
    nanv = np.full(4, np.nan, F)
    out = ensi(blat, blon, nanv, nanv, bg, plat, plon, nanv[:1], nanv[:1], obs, sig, pbg, 10000, 0, 0, 5, True)
    bx, by, bz = convert_coordinates(blat, blon)
    px, py, pz = convert_coordinates(plat, plon)
    yhat = seq_mean(pbg[0])
    yrow = (pbg[0] - yhat).astype(F)
    closed = bg.astype(np.float64).copy()
    for c in range(4):
        rho, _ = corr_barnes((bx[c], by[c], bz[c], F(np.nan), F(np.nan)), (px, py, pz, nanv[:1], nanv[:1]), 10000, 0, 0)
        if rho[0] > 0:
            closed[c] = one_obs_closed_form(bg[c], yrow, rho[0], sig[0], obs[0], yhat)
    err = np.max(np.abs(out - closed) / np.maximum(np.abs(closed), 1e-2))
    print("1-obs closed form vs the LAPACK restatement: max rel err %.2e" % err)
    assert err < 2e-6, err
    assert (out[3] == bg[3]).all()    # out of range: untouched
    for k, val in dict(blat=blat, blon=blon, background=bg, plat=plat, plon=plon, pobs=obs, psigmas=sig, pbackground=pbg,
                       params=np.array([10000, 0, 0, 5, 1.0]), expected=closed).items():
        flat["one_obs_closed_form/" + k] = val
    np.savez_compressed(dst, **flat)
    print("wrote", os.path.normpath(dst), "%.0f KB" % (os.path.getsize(dst) / 1024))


if __name__ == "__main__":
    sys.exit(main())
