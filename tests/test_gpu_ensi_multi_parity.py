"""optimal_interpolation_ensi_multi_{ebe, ebesc, utem} on the GPU vs (a) the golden vectors of the independent numpy / LAPACK
restatement (tests/golden/ensi_multi_cases.npz) and (b) the oracle on further random cases.  Tolerance 1e-5 relative."""
import numpy as np
import pytest

from tests import ensi_multi_golden as G

pytestmark = pytest.mark.gpu


def _run(c, grid_overload):
    import gridpp_amd as gridpp
    h, v, w, mp, allow = c["params"]
    variant = str(c["variant"])
    Y, X, E = [int(t) for t in c["shape"]]
    points = gridpp.Points(c["plat"], c["plon"], c["pelev"], c["plaf"])
    st = gridpp.BarnesStructure(h, v, w)
    if grid_overload:
        b = gridpp.Grid(c["blat"].reshape(Y, X), c["blon"].reshape(Y, X), c["belev"].reshape(Y, X), c["blaf"].reshape(Y, X))
        br, bg, bgc = c["bratios"].reshape(Y, X), c["background"].reshape(Y, X, E), c["background_corr"].reshape(Y, X, E)
    else:
        b = gridpp.Points(c["blat"], c["blon"], c["belev"], c["blaf"])
        br, bg, bgc = c["bratios"], c["background"], c["background_corr"]
    if variant == "ebe":
        out = gridpp.optimal_interpolation_ensi_multi_ebe(b, br, bg, bgc, points, c["pobs"], c["pratios"], c["pbackground"], c["pbackground_corr"], st, int(mp), bool(allow))
    elif variant == "ebesc":
        out = gridpp.optimal_interpolation_ensi_multi_ebesc(b, br, bg, points, c["pobs"], c["pratios"], c["pbackground"], st, int(mp), bool(allow))
    else:
        out = gridpp.optimal_interpolation_ensi_multi_utem(b, br, bg, bgc, points, c["pobs"], c["pratios"], c["pbackground"], c["pbackground_corr"], st, int(mp), bool(allow))
    out = np.asarray(out)
    assert out.dtype == np.float32 and out.shape == bg.shape
    return out.reshape(-1, E)


@pytest.mark.parametrize("name", G.NAMES)
@pytest.mark.parametrize("grid_overload", [False, True])
def test_ensi_multi_golden_vectors(name, grid_overload):
    c = G.CASES[name]
    G.check(_run(c, grid_overload), c)


@pytest.mark.parametrize("variant", ["ebe", "ebesc", "utem"])
def test_ensi_multi_random_vs_oracle(variant):
    from oracle import oracle as O
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    rng = np.random.default_rng(7)
    n, E, S = 300, 12, 70
    blat, blon = rng.random(n).astype(np.float32), rng.random(n).astype(np.float32)
    plat, plon = rng.random(S).astype(np.float32), rng.random(S).astype(np.float32)
    bg, bgc = rng.normal(0, 1, (n, E)).astype(np.float32), rng.normal(0, 1, (n, E)).astype(np.float32)
    pbg, pbgc = rng.normal(0, 1, (S, E)).astype(np.float32), rng.normal(0, 1, (S, E)).astype(np.float32)
    bg[:, E - 1] = np.nan                                   # the last member is invalid: it must stay untouched
    pobs = rng.normal(0, 1, S).astype(np.float32) if variant == "utem" else rng.normal(0, 1, (S, E)).astype(np.float32)
    pr, br = rng.uniform(0.1, 1, S).astype(np.float32), rng.uniform(0.5, 1.5, n).astype(np.float32)
    nanv_b, nanv_p = np.full(n, np.nan, np.float32), np.full(S, np.nan, np.float32)
    c = dict(variant=np.array(variant), shape=np.array([0, n, E]), blat=blat, blon=blon, belev=nanv_b, blaf=nanv_b, bratios=br, background=bg,
             background_corr=bgc, plat=plat, plon=plon, pelev=nanv_p, plaf=nanv_p, pobs=pobs, pratios=pr, pbackground=pbg, pbackground_corr=pbgc,
             params=np.array([25000, 0, 0, 10, 0.0]))
    ref = O.oi_ensi_multi(variant, O.Pts(blat, blon), br, bg, bgc, O.Pts(plat, plon), pobs, pr, pbg, pbgc, O.Barnes(25000), 10, False)
    out = _run(c, False)
    assert np.isnan(out[:, E - 1]).all()
    m = ~np.isnan(ref)
    err = np.abs(out[m].astype(np.float64) - ref[m]) / np.maximum(np.abs(ref[m]), 1e-2)
    assert err.max() < 1e-5, err.max()


@pytest.mark.parametrize("variant,E,S,mp", [("ebe", 10, 140, 100), ("ebesc", 10, 140, 0), ("utem", 80, 120, 0), ("utem", 9, 700, 0)])
def test_ensi_multi_beyond_the_lds_areas(variant, E, S, mp):
    """No capacity limit (oi_ensi_multi.cpp:395-418,489-505 have none): 100 / 140 selected observations for ebe / ebesc (pivoted LU
    in HBM scratch), 80 valid members and 700 selected observations for utem (k_ensi_multi_huge), against the oracle."""
    from oracle import oracle as O
    rng = np.random.default_rng(70 + E + S)
    n = 10
    blat, blon = rng.random(n).astype(np.float32), rng.random(n).astype(np.float32)
    plat, plon = rng.random(S).astype(np.float32), rng.random(S).astype(np.float32)
    bg, bgc = rng.normal(0, 1, (n, E)).astype(np.float32), rng.normal(0, 1, (n, E)).astype(np.float32)
    pbg, pbgc = rng.normal(0, 1, (S, E)).astype(np.float32), rng.normal(0, 1, (S, E)).astype(np.float32)
    pobs = rng.normal(0, 1, S).astype(np.float32) if variant == "utem" else rng.normal(0, 1, (S, E)).astype(np.float32)
    pr, br = rng.uniform(0.5, 1.5, S).astype(np.float32), rng.uniform(0.5, 1.5, n).astype(np.float32)
    nanv_b, nanv_p = np.full(n, np.nan, np.float32), np.full(S, np.nan, np.float32)
    c = dict(variant=np.array(variant), shape=np.array([0, n, E]), blat=blat, blon=blon, belev=nanv_b, blaf=nanv_b, bratios=br, background=bg,
             background_corr=bgc, plat=plat, plon=plon, pelev=nanv_p, plaf=nanv_p, pobs=pobs, pratios=pr, pbackground=pbg, pbackground_corr=pbgc,
             params=np.array([200000, 0, 0, mp, 1.0]))   # every observation in range of every grid point
    ref = O.oi_ensi_multi(variant, O.Pts(blat, blon), br, bg, bgc, O.Pts(plat, plon), pobs, pr, pbg, pbgc, O.Barnes(200000), mp, True)
    out = _run(c, False)
    m = ~np.isnan(ref)
    assert (np.isnan(out) == np.isnan(ref)).all()
    err = np.abs(out[m].astype(np.float64) - ref[m]) / np.maximum(np.abs(ref[m]), 1e-2)
    assert err.max() < 1e-5, err.max()
    assert np.abs(out[m] - bg[m]).max() > 1e-3


def test_ensi_multi_invalid_member_in_front_raises():
    """An invalid member in front of a valid one: the reference indexes lInnov(i, ei) out of bounds (oi_ensi_multi.cpp:565)."""
    c = dict(G.CASES["ebesc_e10_mp8"])
    bg = c["background"].copy()
    bg[0, 0] = np.nan
    c["background"] = bg
    with pytest.raises(RuntimeError, match="out of bounds"):
        _run(c, False)
