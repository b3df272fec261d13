"""Loader of tests/golden/ensi_multi_cases.npz (tools/make_ensi_multi_fixtures.py: independent numpy + scipy.linalg restatement of
src/api/oi_ensi_multi.cpp:329-1311)."""
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ensi_multi_cases.npz")
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
    assert np.nanmax(np.abs(out - case["background"].reshape(out.shape))) > 0.05
