"""Pins oracle/gridpp_oracle.c against the known-answer values of the
reference's own unit tests (tests/golden/reference_known_answers.json)."""
import pytest

from tests import pins, refapi


@pytest.mark.parametrize("pin", pins.ALL_PINS, ids=lambda f: f.__name__)
def test_oracle_pin(pin, golden):
    pin(refapi, golden)


# ---- EnSI: the reference's tests hold no numeric value (tests/test_optimal_interpolation_ens.py:9-35), so the oracle's EnSI
# part (own LU inverse + cyclic Jacobi) is pinned by vectors from an independent LAPACK restatement and a closed form
from tests import ensi_golden  # noqa: E402


@pytest.mark.parametrize("name", ensi_golden.NAMES)
def test_oracle_ensi_golden(name):
    import numpy as np
    from oracle import oracle as O
    c = ensi_golden.CASES[name]
    h, v, w, mp, allow = c["params"]
    nan_b = np.full(c["blat"].size, np.nan, np.float32)
    nan_p = np.full(c["plat"].size, np.nan, np.float32)
    g = O.Pts(c["blat"], c["blon"], c.get("belev", nan_b), c.get("blaf", nan_b))
    p = O.Pts(c["plat"], c["plon"], c.get("pelev", nan_p), c.get("plaf", nan_p))
    if "hfield" in c:   # spatially varying scales on the background grid: the structure as seen from each grid point
        R = np.array([O.structure_localization("Barnes", hh, 0.0013) for hh in c["hfield"]], np.float32)
        out = O.oi_ensi_generic(g, c["background"], p, c["pobs"], c["psigmas"], c["pbackground"], O.Struct("Barnes", h, v, w), int(mp), bool(allow),
                                cell_params=[c["hfield"], c["vfield"], c["wfield"], R])
    else:
        out = O.oi_ensi(g, c["background"], p, c["pobs"], c["psigmas"], c["pbackground"], O.Barnes(h, v, w), int(mp), bool(allow))
    ensi_golden.check(out, c)


# ---- ensi_multi: no test or known answer in the reference; pinned by an independent LAPACK restatement
from tests import ensi_multi_golden  # noqa: E402


@pytest.mark.parametrize("name", ensi_multi_golden.NAMES)
def test_oracle_ensi_multi_golden(name):
    from oracle import oracle as O
    c = ensi_multi_golden.CASES[name]
    h, v, w, mp, allow = c["params"]
    g = O.Pts(c["blat"], c["blon"], c["belev"], c["blaf"])
    p = O.Pts(c["plat"], c["plon"], c["pelev"], c["plaf"])
    out = O.oi_ensi_multi(str(c["variant"]), g, c["bratios"], c["background"], c["background_corr"], p, c["pobs"], c["pratios"],
                          c["pbackground"], c["pbackground_corr"], O.Barnes(h, v, w), int(mp), bool(allow))
    ensi_multi_golden.check(out, c)


# ---- optimal_interpolation_full against the LAPACK golden vectors (tools/make_oi_fixtures.py) --------------------------------------
from tests import oi_golden  # noqa: E402


@pytest.mark.parametrize("name", oi_golden.NAMES)
def test_oi_golden_vectors(name):
    """The oracle's OI (top-max_points cut, its own Cholesky / LU solve, clamp, variance) against an independent numpy + LAPACK
    restatement of src/api/oi.cpp:176-338."""
    from oracle import oracle as O
    c = oi_golden.CASES[name]
    h, v, w, mp, allow = c["params"]
    og = O.Pts(c["blat"], c["blon"], c["belev"], c["blaf"])
    op = O.Pts(c["plat"], c["plon"], c["pelev"], c["plaf"])
    if int(c["kind"]) == 1:
        out, var = O.oi_full_generic(og, c["background"], c["bvariance"], op, c["pobs"], c["obs_variance"], c["pbackground"],
                                     c["bvariance_at_points"], O.Struct("Cressman", h, v, w), int(mp), bool(allow))
    else:
        out, var = O.oi_full(og, c["background"], c["bvariance"], op, c["pobs"], c["obs_variance"], c["pbackground"], c["bvariance_at_points"],
                             O.Barnes(h, v, w), int(mp), bool(allow))
    oi_golden.check(out, var, c)


# ---- a third opinion on the EnSI matrix functions: 50-digit arithmetic ------------------------------------------------------------------
# The reference holds no numeric EnSI value (tests/test_optimal_interpolation_ens.py:9-35), so the oracle is pinned by the numpy / LAPACK
# restatement (tools/make_ensi_fixtures.py).  Both could share a misreading of the SOURCE, but not an arithmetic accident: here the
# restatement's `inv` and `eig_sym` (oi_ensi.cpp:399-421) are replaced by mpmath at 50 digits -- exact for every purpose -- and the
# oracle's float32 outputs must still be the restatement's, also on inputs where the E x E system is ill-conditioned (sigmas x 0.01:
# cond ~ 1e4; round 5 found the large-n KERNELS wrong there while oracle, LAPACK and the 50-digit evaluation agreed to the last bit).
@pytest.mark.parametrize("E,S,mp,sig_scale", [(8, 40, 10, 1.0), (17, 60, 0, 1.0), (12, 50, 40, 0.01), (20, 60, 20, 0.1)])
def test_oracle_ensi_against_50_digit_matrix_functions(E, S, mp, sig_scale):
    mp_ = pytest.importorskip("mpmath")
    import os
    import sys
    import numpy as np
    import scipy.linalg as sla
    from oracle import oracle as O
    from tests.test_gpu_ensi_parity import case
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_ensi_fixtures as M
    mp_.mp.dps = 50

    class Exact:
        lapack = sla.lapack

        @staticmethod
        def inv(A):
            return np.array((mp_.matrix(A.tolist()) ** -1).tolist(), dtype=np.float64)

        @staticmethod
        def eigh(A):
            ev, Q = mp_.eigsy(mp_.matrix(A.tolist()))
            return np.array([float(e) for e in ev]), np.array(Q.tolist(), dtype=np.float64)
    Y, X = 4, 5
    lats, lons, bg, plat, plon, pbg, obs, sig = case(900 + E, Y, X, E, S)
    sig = (sig * sig_scale).astype(np.float32)
    h = 40000.0
    ref = O.oi_ensi(O.Pts(lats.ravel(), lons.ravel()), bg.reshape(-1, E), O.Pts(plat, plon), obs, sig, pbg, O.Barnes(h), mp, True)
    nanb, nanp = np.full(Y * X, np.nan, np.float32), np.full(S, np.nan, np.float32)
    saved = M.sla
    try:
        M.sla = Exact
        exact = M.ensi(lats.ravel().astype(np.float32), lons.ravel().astype(np.float32), nanb, nanb, bg.reshape(-1, E), plat.astype(np.float32),
                       plon.astype(np.float32), nanp, nanp, obs, sig, pbg, h, 0.0, 0.0, mp, True)
    finally:
        M.sla = saved
    m = ~np.isnan(exact)
    assert (np.isnan(ref) == np.isnan(exact)).all() and m.any()
    err = np.abs(ref[m].astype(np.float64) - exact[m]) / np.maximum(np.abs(exact[m]), 1e-2)
    assert err.max() < 1e-6, err.max()                                   # (in practice 0: the same float32 on every value)
    assert (ref[m] != exact[m]).mean() < 0.01
