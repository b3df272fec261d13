"""Randomised parity soak of neighbourhood_quantile_fast (3-D input) vs the oracle, aimed at the round-3 kernels: the rank /
sum-of-absolute-differences count pass (k_qf_lut, k_qf_count) and the marching box pass (k_qf_box<HW, flagged?>): member counts
with and without whole float4 rows, 1..16 thresholds in ascending / random order, with duplicates, clustered (two in one bucket:
the compare-per-threshold pass), with non-finite ones; halfwidths 0..16 (and beyond: the unfused path); several 256-column
strips and 64-row segments; missing members (flagged rows), cells without any valid member, scalar and field quantiles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gridpp_amd as gridpp
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t0, seed, bad, kinds = time.time(), 0, [], {}
while time.time() - t0 < budget:
    seed += 1
    rng = np.random.default_rng(9000 + seed)
    big = seed % 5 == 0
    Y = int(rng.integers(60, 100)) if big else int(rng.integers(1, 70))
    X = int(rng.choice([260, 300, 515])) if big else int(rng.integers(1, 90))
    E = int(rng.choice([4, 8, 12, 20, 100, 3, 10, 1]))
    if big:
        E = int(rng.choice([4, 8, 5]))
    hw = int(rng.choice([0, 1, 2, 3, 5, 7, 8, 15, 16, 20]))
    if big:
        hw = min(hw, 8)   # (the oracle's box sums are what takes the time)
    T = int(rng.integers(1, 17))
    kind = int(rng.integers(0, 6))
    thr = np.sort(rng.uniform(-2, 12, T)).astype(np.float32)
    if kind == 1:
        thr = rng.permutation(thr)
    elif kind == 2 and T > 1:
        thr[rng.integers(0, T)] = thr[rng.integers(0, T)]              # a duplicate
    elif kind == 3 and T > 1:
        thr[1] = np.nextafter(thr[0], np.float32(np.inf))               # two thresholds one ulp apart: one bucket
    elif kind == 4:
        thr[rng.integers(0, T)] = rng.choice([np.inf, -np.inf, np.nan])
    elif kind == 5:
        thr = np.round(thr).astype(np.float32)                          # integers: members land exactly on thresholds
    f = rng.uniform(0, 10, (Y, X, E)).astype(np.float32)
    if kind == 5:
        f = np.round(f).astype(np.float32)
    mode = seed % 4
    if mode == 1:
        f[rng.random(f.shape) < 0.05] = np.nan
    elif mode == 2:
        f[Y // 2:Y // 2 + 3, X // 3:X // 3 + 5, :] = np.nan            # cells without a valid member
        f[rng.random(f.shape) < 0.01] = np.inf
        f[rng.random(f.shape) < 0.01] = -np.inf
    if seed % 3 == 0:
        q = rng.random((Y, X)).astype(np.float32)
        q[0, 0] = 0.0
        q[-1, -1] = 1.0
        if Y * X > 4:
            q[Y // 2, X // 2] = np.nan
    else:
        q = float(rng.choice([0.0, 0.25, 0.5, 0.9, 1.0]))
    try:
        out = np.asarray(gridpp.neighbourhood_quantile_fast(f, q, hw, thr))
        ref = O.neighbourhood_quantile_fast(f, q if isinstance(q, np.ndarray) else [q], hw, thr)
        assert out.shape == ref.shape and (np.isnan(out) == np.isnan(ref)).all(), "NaN pattern"
        m = ~np.isnan(ref) & ~np.isinf(ref)
        assert (out[np.isinf(ref)] == ref[np.isinf(ref)]).all()
        if m.any():
            err = np.abs(out[m].astype(np.float64) - ref[m]) / np.maximum(np.abs(ref[m]), 1e-3)
            assert err.max() < 1e-5, err.max()
        kinds[kind] = kinds.get(kind, 0) + 1
    except AssertionError as e:
        bad.append((seed, Y, X, E, hw, T, kind, mode, str(e)[:80]))
print("seeds: %d, failures: %d, threshold kinds (0 ascending, 1 permuted, 2 duplicate, 3 clustered, 4 non-finite, 5 integer ties): %s" % (seed, len(bad), sorted(kinds.items())))
for b in bad[:10]:
    print(b)
sys.exit(1 if bad else 0)
