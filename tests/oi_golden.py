"""Loader of tests/golden/oi_cases.npz (made by tools/make_oi_fixtures.py in the build container: an independent numpy +
scipy.linalg/LAPACK restatement of src/api/oi.cpp:176-338 -- top-max_points cut, multi-observation inverse, anti-extrapolation clamp,
analysis variance)."""
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oi_cases.npz")
RTOL = 1e-5


def load():
    z = np.load(_PATH)
    cases = {}
    for key in z.files:
        name, field = key.split("/")
        cases.setdefault(name, {})[field] = z[key]
    return cases


CASES = load()
NAMES = sorted(CASES)


def check(out, var, case):
    for got, exp, floor in ((out, case["expected"], 1e-2), (var, case["expected_variance"], 1e-3)):
        got = np.asarray(got).ravel()
        exp = exp.ravel()
        assert got.dtype == np.float32
        assert (np.isnan(got) == np.isnan(exp)).all()
        m = ~np.isnan(exp)
        err = np.abs(got[m].astype(np.float64) - exp[m].astype(np.float64)) / np.maximum(np.abs(exp[m]), floor)
        assert err.max() < RTOL, err.max()
    bg = case["background"].ravel()
    assert np.nanmax(np.abs(np.asarray(out).ravel() - bg)) > 0.05      # the analysis actually moved
