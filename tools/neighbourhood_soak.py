"""Randomised neighbourhood parity soak vs the oracle: 2-D and 3-D fields of random shape, halfwidth, missing-value patterns;
Mean / Sum / Count / Min / Max (neighbourhood), exact quantile, quantile_fast.  Tolerances as tests/test_gpu_neighbourhood_parity.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gridpp_amd as gridpp
from oracle import oracle as O


def close(a, b, exact=False, scale=1e-3):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == np.float32, (a.shape, b.shape)
    assert (np.isnan(a) == np.isnan(b)).all(), "NaN pattern"
    m = ~np.isnan(b) & ~np.isinf(b)
    assert (a[np.isinf(b)] == b[np.isinf(b)]).all()
    if not m.any():
        return
    if exact:
        assert (a[m] == b[m]).all()
    else:   # sums are compared against the magnitude of what was summed (a window sum of mixed signs can be ~0)
        err = np.abs(a[m].astype(np.float64) - b[m]) / np.maximum(np.abs(b[m]), scale)
        assert err.max() < 1e-5, err.max()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t0, seed, bad = time.time(), 0, []
while time.time() - t0 < budget:
    seed += 1
    rng = np.random.default_rng(seed)
    Y, X = int(rng.integers(1, 90)), int(rng.integers(1, 120))
    E = int(rng.choice([0, 0, 1, 3, 20, 100, 161]))
    hw = int(rng.choice([0, 1, 2, 5, 15, 40, 200]))
    shape = (Y, X) if E == 0 else (Y, X, E)
    f = rng.uniform(-5, 10, shape).astype(np.float32)
    mode = seed % 4
    if mode == 1:
        f[rng.random(shape) < 0.1] = np.nan
    elif mode == 2:
        f[: max(1, Y // 3)] = np.nan
        f[rng.random(shape) < 0.02] = np.inf
    elif mode == 3:
        f[...] = np.nan if seed % 8 == 3 else f
    try:
        for stat in (gridpp.Mean, gridpp.Sum):
            close(gridpp.neighbourhood(f, hw, stat), O.neighbourhood(f, hw, stat), scale=(1.0 if stat == gridpp.Mean else 10.0 * min((2 * hw + 1) ** 2, Y * X)))
        for stat in (gridpp.Count, gridpp.Min, gridpp.Max):
            close(gridpp.neighbourhood(f, hw, stat), O.neighbourhood(f, hw, stat), exact=True)
        if Y * X * max(E, 1) * (2 * min(hw, 8) + 1) ** 2 < 3e7:
            h2 = min(hw, 8)
            q = float(rng.choice([0.0, 0.1, 0.5, 0.9, 1.0]))
            close(gridpp.neighbourhood_quantile(f, q, h2), O.neighbourhood_quantile(f, q, h2))
        T = int(rng.choice([1, 2, 7, 30]))
        thr = np.sort(rng.uniform(-5, 10, T)).astype(np.float32)
        qq = float(rng.choice([0.0, 0.3, 0.5, 1.0]))
        close(gridpp.neighbourhood_quantile_fast(f, qq, hw, thr), O.neighbourhood_quantile_fast(f, [qq], hw, thr))
    except AssertionError as e:
        bad.append((seed, Y, X, E, hw, mode, str(e)[:100]))
print("seeds: %d, failures: %d" % (seed, len(bad)))
for b in bad[:10]:
    print(b)
