"""Loader of tests/golden/ensi_cases.npz (made by tools/make_ensi_fixtures.py in the build container: an independent
numpy + scipy.linalg/LAPACK restatement of src/api/oi_ensi.cpp:114-568, plus one analytic 1-observation case)."""
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ensi_cases.npz")
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


def check(out, case):
    exp = case["expected"].reshape(out.shape)
    assert (np.isnan(out) == np.isnan(exp)).all()
    m = ~np.isnan(exp)
    err = np.abs(out[m].astype(np.float64) - exp[m].astype(np.float64)) / np.maximum(np.abs(exp[m]), 1e-2)
    assert err.max() < RTOL, err.max()
    bg = case["background"].reshape(out.shape)
    assert np.nanmax(np.abs(out - bg)) > 0.05
