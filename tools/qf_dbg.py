import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import gridpp_amd as gridpp
from oracle import oracle as O
def run(seed):
    rng = np.random.default_rng(9000 + seed)
    big = seed % 5 == 0
    Y = int(rng.integers(60, 150)) if big else int(rng.integers(1, 70))
    X = int(rng.choice([260, 300, 512, 515])) if big else int(rng.integers(1, 90))
    E = int(rng.choice([4, 8, 12, 20, 100, 3, 10, 1]))
    if big:
        E = int(rng.choice([4, 8, 5]))
    hw = int(rng.choice([0, 1, 2, 3, 5, 7, 8, 15, 16, 20]))
    T = int(rng.integers(1, 17))
    kind = int(rng.integers(0, 6))
    thr = np.sort(rng.uniform(-2, 12, T)).astype(np.float32)
    if kind == 1: thr = rng.permutation(thr)
    elif kind == 2 and T > 1: thr[rng.integers(0, T)] = thr[rng.integers(0, T)]
    elif kind == 3 and T > 1: thr[1] = np.nextafter(thr[0], np.float32(np.inf))
    elif kind == 4: thr[rng.integers(0, T)] = rng.choice([np.inf, -np.inf, np.nan])
    elif kind == 5: thr = np.round(thr).astype(np.float32)
    f = rng.uniform(0, 10, (Y, X, E)).astype(np.float32)
    if kind == 5: f = np.round(f).astype(np.float32)
    mode = seed % 4
    if mode == 1: f[rng.random(f.shape) < 0.05] = np.nan
    elif mode == 2:
        f[Y // 2:Y // 2 + 3, X // 3:X // 3 + 5, :] = np.nan
        f[rng.random(f.shape) < 0.01] = np.inf
        f[rng.random(f.shape) < 0.01] = -np.inf
    if seed % 3 == 0:
        q = rng.random((Y, X)).astype(np.float32); q[0, 0] = 0.0; q[-1, -1] = 1.0
        if Y * X > 4: q[Y // 2, X // 2] = np.nan
    else:
        q = float(rng.choice([0.0, 0.25, 0.5, 0.9, 1.0]))
    out = np.asarray(gridpp.neighbourhood_quantile_fast(f, q, hw, thr))
    ref = O.neighbourhood_quantile_fast(f, q if isinstance(q, np.ndarray) else [q], hw, thr)
    os.environ["GPP_QF_NO_RANKS"] = "1"
    out2 = np.asarray(gridpp.neighbourhood_quantile_fast(f, q, hw, thr))
    del os.environ["GPP_QF_NO_RANKS"]
    os.environ["GPP_QF_NO_FUSED"] = "1"
    out3 = np.asarray(gridpp.neighbourhood_quantile_fast(f, q, hw, thr))
    del os.environ["GPP_QF_NO_FUSED"]
    def cmp(o):
        nanbad = int((np.isnan(o) != np.isnan(ref)).sum())
        m = ~np.isnan(ref) & ~np.isnan(o)
        err = np.abs(o[m].astype(np.float64) - ref[m]) / np.maximum(np.abs(ref[m]), 1e-3) if m.any() else np.zeros(1)
        bad = np.argwhere((np.abs(o.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-3)) > 1e-5)
        return nanbad, float(err.max()), int((err > 1e-5).sum()), bad[:4].tolist()
    print("seed", seed, "Y,X,E,hw,T,kind,mode", Y, X, E, hw, T, kind, mode, "thr", thr.tolist(), "q", q if not isinstance(q, np.ndarray) else "field")
    print("  ranked+box:", cmp(out)); print("  compare-pass+box:", cmp(out2)); print("  unfused:", cmp(out3))
for sd in [int(a) for a in sys.argv[1:]]:
    run(sd)
