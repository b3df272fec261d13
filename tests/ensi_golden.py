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


def rel_err(out, ref, background):
    """|out - ref| relative to max(|ref|, 1e-2, one float32 ulp of the operands).  An EnSI output is the float32 sum
    `ensemble mean + increment` (oi_ensi.cpp:553) of two terms of the size of the member values, so one ulp of THOSE (2^-23 of the
    largest member of the cell) is the resolution of the reference's own arithmetic however small the sum comes out; 1e-5 relative is
    asked against that resolution, not below it (a last-bit difference of a term near 1 against a result of 0.0099 is 1.2e-5 of the
    result)."""
    out, ref, background = np.asarray(out), np.asarray(ref), np.asarray(background)
    scale = np.nanmax(np.abs(np.where(np.isfinite(background), background, np.nan).reshape(ref.shape)), axis=-1, keepdims=True)
    scale = np.where(np.isfinite(scale), scale, 0.0)
    floor = np.maximum(1e-2, 1.2e-7 * scale / 1e-5)          # abs tolerance 1e-5 * floor = one ulp of the largest member
    m = ~np.isnan(ref)
    den = np.maximum(np.abs(ref), np.broadcast_to(floor, ref.shape))
    return (np.abs(out.astype(np.float64) - ref) / den)[m]


def check(out, case):
    exp = case["expected"].reshape(out.shape)
    assert (np.isnan(out) == np.isnan(exp)).all()
    err = rel_err(out, exp.astype(np.float64), case["background"])
    assert err.max() < RTOL, err.max()
    bg = case["background"].reshape(out.shape)
    assert np.nanmax(np.abs(out - bg)) > 0.05
