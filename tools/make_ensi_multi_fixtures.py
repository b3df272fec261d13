#!/usr/bin/env python
"""Golden vectors for optimal_interpolation_ensi_multi_{ebe,ebesc,utem} (build container only).

Independent numpy + scipy.linalg (LAPACK) restatement of /root/reference/src/api/oi_ensi_multi.cpp:329-1311 (Points overloads;
the Grid overloads, :34-327, flatten row-major and delegate).  The reference holds no test or known answer for these functions,
so these vectors pin the oracle's restatement (oracle/gridpp_oracle.c) and, through it, the HIP kernels.  Shares the geometry /
structure helpers with tools/make_ensi_fixtures.py (which are themselves independent of the oracle).

All members are valid in every case: with an invalid member in front of a valid one the reference indexes `lInnov(i, ei)` with
the ORIGINAL member index into a matrix that has only nValidEns columns (oi_ensi_multi.cpp:563-567,770-773) -- out of bounds.

    python tools/make_ensi_multi_fixtures.py   ->  tests/golden/ensi_multi_cases.npz
"""
import os
import sys

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_ensi_fixtures import F, convert_coordinates, corr_barnes, get_neighbours, localization_distance, seq_mean  # noqa: E402

MIN_STD = F(0.0013)


def seq_std(row):
    """calc_statistic(Std), util.cpp:40-75: shifted by the first valid value, float accumulation."""
    ok = np.isfinite(row)
    if not ok.any():
        return F(np.nan)
    v = row[ok].astype(F)
    K = v[0]
    d = (v - K).astype(F)
    total = np.cumsum(d, dtype=F)[-1]
    total2 = np.cumsum((d * d).astype(F), dtype=F)[-1]
    mean, mean2 = F(total / F(v.size)), F(total2 / F(v.size))
    var = F(mean2 - F(mean * mean))
    if var < 0:
        var = F(0)
    return F(np.sqrt(var))


def select(y, bx, by, bz, belev, blaf, px, py, pz, pelev, plaf, obs_valid, h, v, w, max_points, loc):
    """oi_ensi_multi.cpp:459-515: candidates in range, valid first-member observation, rho > 0, best max_points by rho."""
    idx0 = get_neighbours(px, py, pz, bx[y], by[y], bz[y], loc)
    if idx0.size == 0:
        return idx0, None
    p1 = (bx[y], by[y], bz[y], F(belev[y]), F(blaf[y]))
    rhos, _ = corr_barnes(p1, (px[idx0], py[idx0], pz[idx0], pelev[idx0], plaf[idx0]), h, v, w)
    keep = obs_valid[idx0] & (rhos > 0)
    cand, crho = idx0[keep], rhos[keep]
    if max_points > 0 and cand.size > max_points:
        order = np.argsort(crho, kind="stable")[::-1]
        assert crho[order[max_points - 1]] != crho[order[max_points]], "rho tie at the cut"
        return cand[order[:max_points]], crho[order[:max_points]].astype(np.float64)
    return cand, crho.astype(np.float64)


def clamp(dx, lInnov):
    """oi_ensi_multi.cpp:588-615 (ebe / ebesc anti-extrapolation)."""
    mx, mn = lInnov.max(axis=0), lInnov.min(axis=0)
    out = dx.copy()
    for e in range(dx.size):
        inc, a, b = F(dx[e]), F(mx[e]), F(mn[e])
        if a > 0 and inc > a:
            inc = a
        elif a < 0 and inc > 0:
            inc = F(0)
        elif b < 0 and inc < b:
            inc = b
        elif b > 0 and inc < 0:
            inc = F(0)
        out[e] = inc
    return out


def obs_corr(sel, px, py, pz, pelev, plaf, h, v, w):
    lS = sel.size
    L = np.empty((lS, lS))
    for i in range(lS):
        s = sel[i]
        c, _ = corr_barnes((px[s], py[s], pz[s], pelev[s], plaf[s]), (px[sel], py[sel], pz[sel], pelev[sel], plaf[sel]), h, v, w)
        L[i] = c.astype(np.float64)
    return L


def ensi_multi(variant, blat, blon, belev, blaf, bratios, background, background_corr, plat, plon, pelev, plaf, pobs, pratios,
               pbackground, pbackground_corr, h, v, w, max_points, allow):
    background = background.astype(F)
    nY, nE = background.shape
    nS = plat.size
    out = background.copy()
    if nS == 0:
        return out
    bx, by, bz = convert_coordinates(blat, blon)
    px, py, pz = convert_coordinates(plat, plon)
    pelev, plaf = pelev.astype(F), plaf.astype(F)
    fields = [background, pbackground] + ([background_corr, pbackground_corr] if variant != "ebesc" else [])
    valid = [e for e in range(nE) if all(np.isfinite(f[:, e]).all() for f in fields)]
    nV = len(valid)
    if nV == 0:
        return out
    assert valid == list(range(nV)), "an invalid member in front of a valid one: out-of-bounds indexing in the reference"
    loc = localization_distance(h)
    obs_valid = np.isfinite(pobs[:, 0]) if variant != "utem" else np.isfinite(pobs)
    if variant == "ebe":
        gZ = np.zeros((nS, nV), F)                                                    # :421-444
        for i in range(nS):
            r = pbackground_corr[i, valid].astype(F)
            mean, std = seq_mean(r), seq_std(r)
            if np.isfinite(mean) and np.isfinite(std) and std > MIN_STD:
                gZ[i] = (1 / np.sqrt(np.float64(nV - 1)) * (r - mean).astype(F).astype(np.float64) / np.float64(std)).astype(F)
    if variant == "utem":
        const_fact = F(1 / np.sqrt(np.float64(nV - 1)))                              # :968 (float)
        gY, gYc, gYhat = np.zeros((nS, nV), F), np.zeros((nS, nV), F), np.zeros(nS, F)
        for i in range(nS):                                                          # :969-1003
            r, rc = pbackground[i, valid].astype(F), pbackground_corr[i, valid].astype(F)
            mean = seq_mean(r)
            if np.isfinite(mean):
                gY[i] = (r - mean).astype(F)
            gYhat[i] = mean
            mc, sc = seq_mean(rc), seq_std(rc)
            if np.isfinite(mc) and np.isfinite(sc) and sc > MIN_STD:
                gYc[i] = ((const_fact * (rc - mc).astype(F)).astype(F) / sc).astype(F)
    for y in range(nY):
        ratio = F(bratios[y])
        sel, lRhos = select(y, bx, by, bz, belev, blaf, px, py, pz, pelev, plaf, obs_valid, h, v, w, max_points, loc)
        if sel.size == 0:
            continue
        lS = sel.size
        if variant in ("ebe", "ebesc"):
            lInnov = (pobs[sel][:, valid].astype(F) - pbackground[sel][:, valid].astype(F)).astype(F).astype(np.float64)   # :565 (float difference)
            Rdd = np.diag(pratios[sel].astype(F).astype(np.float64))
            L2 = obs_corr(sel, px, py, pz, pelev, plaf, h, v, w)
            if variant == "ebe":
                r = background_corr[y, valid].astype(F)                              # :531-541
                mean, std = seq_mean(r), seq_std(r)
                xL = np.zeros(nV)
                if np.isfinite(mean) and np.isfinite(std) and std > MIN_STD:
                    xL = 1 / np.sqrt(np.float64(nV - 1)) * (r - mean).astype(F).astype(np.float64) / np.float64(std)
                Z = gZ[sel].astype(np.float64)
                r_lr = lRhos * (xL @ Z.T)                                            # :581
                R_rr = L2 * (Z @ Z.T)                                                # :584
                K = r_lr @ sla.inv(R_rr + Rdd)                                       # :586
            else:
                K = lRhos @ sla.inv(L2 + Rdd)                                        # :786
            dx = np.float64(ratio) * (K @ lInnov)                                    # :588 / :788
            if not allow:
                dx = clamp(dx, lInnov)
            out[y, valid] = (background[y, valid].astype(np.float64) + dx).astype(F)   # float + double
        else:
            lObs = pobs[sel].astype(F).astype(np.float64)
            lY, lYc, lYhat = gY[sel].astype(np.float64), gYc[sel].astype(np.float64), gYhat[sel].astype(np.float64)
            Rinv = np.diag(lRhos / pratios[sel].astype(F).astype(np.float64))        # :1075-1079
            C = lYc.T @ Rinv
            Pinv = C @ lYc + np.eye(nV)                                              # :1085
            lu, piv, _ = sla.lapack.dgetrf(Pinv)
            if F(sla.lapack.dgecon(lu, np.linalg.norm(Pinv, 1))[0]) <= 0:
                continue
            P = sla.inv(Pinv)
            ev, evec = sla.eigh(np.float64(nV - 1) * P)                              # :1098
            W = evec @ np.diag(np.sqrt(ev)) @ evec.T
            wv = (P @ C) @ (lObs - lYhat)                                            # :1131-1139
            vals = background[y, valid].astype(F)
            valc = background_corr[y, valid].astype(F)
            ensMean = F(np.cumsum(vals, dtype=F)[-1] / F(nV))
            ensStd = seq_std(vals)
            ensMeanC = F(np.cumsum(valc, dtype=F)[-1] / F(nV))
            ensStdC = seq_std(valc)
            X = vals.astype(np.float64) - np.float64(ensMean)
            if ensStdC <= MIN_STD:
                Xc = np.zeros(nV)
            else:
                Xc = (((const_fact * (valc - ensMeanC).astype(F)).astype(F)) / ensStdC).astype(F).astype(np.float64)   # :1176 float expression into a double vector
            W = np.float64(ensStd) * W + np.float64(ratio) * wv[:, None]            # :1181-1185
            total = np.zeros(nV, F)
            for k in range(nV):                                                      # :1229-1233
                total = (total.astype(np.float64) + Xc[k] * W[k, :]).astype(F)
            curr = total.copy()
            if not allow:                                                            # :1238-1262
                lYflat = lY.flatten(order="F")
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
            out[y, valid] = (ensMean + curr).astype(F)
    return out


def make_case(variant, seed, n, E, S, h, max_points, allow, v=0.0, elev=False, nan_obs=False):
    rng = np.random.default_rng(seed)
    side = int(np.sqrt(n))
    lats, lons = np.meshgrid(np.linspace(0, 1, side), np.linspace(0, 1, side), indexing="ij")
    blat, blon = lats.ravel().astype(F), lons.ravel().astype(F)
    n = blat.size
    base = np.sin(5 * blat) * np.cos(3 * blon)
    bg = (base[:, None] + rng.normal(0, 1, (n, E))).astype(F)
    bgc = (base[:, None] * 0.5 + rng.normal(0, 1, (n, E))).astype(F)
    belev = rng.uniform(0, 500, n).astype(F) if elev else np.full(n, np.nan, F)
    blaf = np.full(n, np.nan, F)
    plat, plon = rng.random(S).astype(F), rng.random(S).astype(F)
    pelev = rng.uniform(0, 500, S).astype(F) if elev else np.full(S, np.nan, F)
    plaf = np.full(S, np.nan, F)
    pbg = rng.normal(0, 1, (S, E)).astype(F)
    pbgc = (pbg * 0.7 + rng.normal(0, 0.7, (S, E))).astype(F)
    if variant == "utem":
        pobs = rng.normal(0, 1, S).astype(F)
        if nan_obs:
            pobs[::7] = np.nan
    else:
        pobs = (rng.normal(0, 1, S)[:, None] + rng.normal(0, 0.3, (S, E))).astype(F)
        if nan_obs:
            pobs[::7, :] = np.nan
    pratios = rng.uniform(0.1, 1, S).astype(F)
    bratios = rng.uniform(0.5, 1.5, n).astype(F)
    out = ensi_multi(variant, blat, blon, belev, blaf, bratios, bg, bgc, plat, plon, pelev, plaf, pobs, pratios, pbg, pbgc, h, v, 0.0,
                     max_points, allow)
    return dict(variant=np.array(variant), shape=np.array([side, side, E]), blat=blat, blon=blon, belev=belev, blaf=blaf, bratios=bratios,
                background=bg, background_corr=bgc, plat=plat, plon=plon, pelev=pelev, plaf=plaf, pobs=pobs, pratios=pratios,
                pbackground=pbg, pbackground_corr=pbgc, params=np.array([h, v, 0.0, max_points, 1.0 if allow else 0.0]), expected=out)


CASES = {
    "ebe_e10_mp8":          dict(variant="ebe", seed=41, n=400, E=10, S=60, h=30000, max_points=8, allow=True),
    "ebe_e20_mp20_noextrap": dict(variant="ebe", seed=42, n=256, E=20, S=80, h=35000, max_points=20, allow=False, nan_obs=True),
    "ebe_e6_mp0_elev":      dict(variant="ebe", seed=43, n=196, E=6, S=40, h=15000, max_points=0, allow=True, v=300.0, elev=True),
    "ebesc_e10_mp8":        dict(variant="ebesc", seed=44, n=400, E=10, S=60, h=30000, max_points=8, allow=True),
    "ebesc_e30_mp25_noextrap": dict(variant="ebesc", seed=45, n=256, E=30, S=90, h=35000, max_points=25, allow=False, nan_obs=True),
    "ebesc_e5_mp0_elev":    dict(variant="ebesc", seed=46, n=196, E=5, S=40, h=15000, max_points=0, allow=True, v=300.0, elev=True),
    "utem_e10_mp8":         dict(variant="utem", seed=47, n=400, E=10, S=60, h=30000, max_points=8, allow=True),
    "utem_e30_mp20_noextrap": dict(variant="utem", seed=48, n=196, E=30, S=90, h=40000, max_points=20, allow=False, nan_obs=True),
    "utem_e6_mp0_elev":     dict(variant="utem", seed=49, n=196, E=6, S=40, h=15000, max_points=0, allow=True, v=300.0, elev=True),
}


def main():
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ensi_multi_cases.npz")
    flat = {}
    for name, kw in CASES.items():
        d = make_case(**kw)
        print("%-28s cells %4d E %2d  max |analysis - background| = %.3f" % (name, d["blat"].size, d["background"].shape[1],
                                                                           np.nanmax(np.abs(d["expected"] - d["background"]))), flush=True)
        for k, val in d.items():
            flat[name + "/" + k] = val
    np.savez_compressed(dst, **flat)
    print("wrote", os.path.normpath(dst), "%.0f KB" % (os.path.getsize(dst) / 1024))


if __name__ == "__main__":
    sys.exit(main())
