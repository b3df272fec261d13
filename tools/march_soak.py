"""Randomised soak of the marching neighbourhood kernels (k_box_march, k_minmax_march: halfwidth <= 16) against the two-pass kernels they replace
(GPP_BOX_TWO_PASS, themselves held to the oracle by tests/ and tools/neighbourhood_soak.py): fields of up to 1500 x 1500 cells (2-D, device resident),
every halfwidth 0 .. 16 (Min / Max: 0 .. 32), Mean / Sum / Count / Min / Max, missing values scattered, in bands of rows, in blocks, infinite values, fields without a
valid value; the launch geometry varied with GPP_BM_FILL (one row segment per strip .. segments of one chunk).  Count, Min and Max must agree bit
for bit, Mean and Sum to 1e-6 (the same doubles added in another order).   python tools/march_soak.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gridpp_amd as gridpp

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t0, seed, bad, ncalls = time.time(), 0, [], 0
stats = [gridpp.Mean, gridpp.Sum, gridpp.Count, gridpp.Min, gridpp.Max]
while time.time() - t0 < budget:
    seed += 1
    rng = np.random.default_rng(9000 + seed)
    big = seed % 5 == 0
    Y, X = (int(rng.integers(200, 1500)), int(rng.integers(200, 1500))) if big else (int(rng.integers(1, 200)), int(rng.integers(1, 300)))
    f = rng.uniform(-5, 10, (Y, X)).astype(np.float32)
    mode = seed % 6
    if mode == 1:
        f[rng.random((Y, X)) < 0.05] = np.nan
    elif mode == 2:
        for y0 in range(int(rng.integers(0, 40)), Y, int(rng.integers(20, 120))):
            f[y0:y0 + int(rng.integers(1, 40))] = np.nan
    elif mode == 3:
        y0, x0 = int(rng.integers(0, Y)), int(rng.integers(0, X))
        f[y0:y0 + 50, x0:x0 + 70] = np.nan
        f[rng.random((Y, X)) < 0.001] = np.inf
    elif mode == 4:
        f[rng.random((Y, X)) < 0.6] = np.nan
    elif mode == 5 and seed % 12 == 5:
        f[...] = np.nan
    d = torch.from_numpy(f).cuda()
    fill = str(int(rng.choice([1, 64, 768, 4096])))
    for hw in ([int(rng.integers(0, 33))] if big else [int(h) for h in rng.choice(33, 3, replace=False)]):
        for stat in (stats if hw <= 16 else [gridpp.Min, gridpp.Max]):     # (17 .. 32: k_minmax_march over strips of 32 columns)
            gridpp.set_path_override("GPP_BM_FILL", fill)
            a = gridpp.neighbourhood(d, hw, stat).cpu().numpy()
            gridpp.set_path_override("GPP_BM_FILL", None)
            gridpp.set_path_override("GPP_BOX_TWO_PASS", "1")
            b = gridpp.neighbourhood(d, hw, stat).cpu().numpy()
            gridpp.set_path_override("GPP_BOX_TWO_PASS", None)
            ncalls += 2
            ok = (np.isnan(a) == np.isnan(b)).all()
            m = ~np.isnan(b)
            if ok and m.any():
                if stat in (gridpp.Count, gridpp.Min, gridpp.Max):
                    ok = (a[m] == b[m]).all()
                else:
                    scale = 1e-3 if stat == gridpp.Mean else 10.0 * min((2 * hw + 1) ** 2, Y * X)
                    ok = (np.abs(a[m].astype(np.float64) - b[m]) <= 1e-6 * np.maximum(np.abs(b[m]), scale)).all()
            if not ok:
                bad.append((seed, Y, X, hw, int(stat), mode, fill))
print("seeds: %d, calls: %d, failures: %d" % (seed, ncalls, len(bad)))
for b in bad[:10]:
    print(b)
sys.exit(1 if bad else 0)
