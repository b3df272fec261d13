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
