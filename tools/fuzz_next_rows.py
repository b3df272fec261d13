"""Randomised comparison of the §8f functions against the oracle over many seeds and shapes (a longer-running companion of
tests/test_gpu_{bilinear,radius,gridops}_parity.py; prints one line per function with the worst deviation seen)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gridpp_amd as gridpp
from oracle import oracle as O
from tests import refapi

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
worst = {}


def note(name, out, ref, exact, scale=1e-3):
    """scale: magnitude of the quantities that were summed (a mean of values near +-2 that comes out as 0.001 is accurate to
    1e-5 of 2, not of 0.001)"""
    out, ref = np.asarray(out, np.float64), np.asarray(ref, np.float64)
    assert out.shape == ref.shape, (name, out.shape, ref.shape)
    assert np.array_equal(np.isnan(out), np.isnan(ref)), name + ": NaN pattern differs"
    m = ~np.isnan(ref)
    err = float(np.max(np.abs(out[m] - ref[m]) / np.maximum(np.abs(ref[m]), scale))) if m.any() else 0.0
    if exact:
        assert err == 0.0, (name, err)
    else:
        assert err < 1e-5, (name, err)
    worst[name] = max(worst.get(name, 0.0), err)


t0, seed = time.time(), 0
while time.time() - t0 < budget:
    seed += 1
    rng = np.random.default_rng(seed)
    ct = int(rng.integers(0, 2))
    Y, X = int(rng.integers(2, 60)), int(rng.integers(2, 70))
    if ct:
        lats, lons = np.meshgrid(np.linspace(0, rng.uniform(2e3, 8e4), Y), np.linspace(-500, rng.uniform(2e3, 9e4), X), indexing="ij")
        jit = 0.2 * (lats[-1, 0] - lats[0, 0]) / max(Y - 1, 1)
    else:
        lats, lons = np.meshgrid(np.linspace(55, 55 + rng.uniform(0.05, 3), Y), np.linspace(5, 5 + rng.uniform(0.05, 4), X), indexing="ij")
        jit = 0.2 * (lats[-1, 0] - lats[0, 0]) / max(Y - 1, 1)
    lats = lats + rng.uniform(-jit, jit, lats.shape)          # mildly irregular mesh: general quadrilaterals
    lons = lons + rng.uniform(-jit, jit, lons.shape)
    elev = rng.uniform(0, 500, (Y, X)).astype(np.float32)
    grid, og = gridpp.Grid(lats, lons, elev, 0 * elev, ct), O.Pts(lats.ravel(), lons.ravel(), elev.ravel(), None, ct)
    n = int(rng.integers(1, 400))
    plat = lats.min() + (lats.max() - lats.min()) * (1.3 * rng.random(n) - 0.15)
    plon = lons.min() + (lons.max() - lons.min()) * (1.3 * rng.random(n) - 0.15)
    pelev = rng.uniform(0, 500, n).astype(np.float32)
    pts, op = gridpp.Points(plat, plon, pelev, 0 * pelev, ct), O.Pts(plat, plon, pelev, None, ct)
    field = rng.normal(0, 2, (Y, X)).astype(np.float32)
    field[rng.random((Y, X)) < rng.choice([0.0, 0.1])] = np.nan
    vals = rng.normal(0, 2, n).astype(np.float32)
    span = float(np.hypot(*(og.x.max() - og.x.min(), og.y.max() - og.y.min())) + 1.0)
    try:
        ref = O.bilinear(og, (Y, X), op, field)
    except O.OracleDistorted:
        ref = None
    if ref is not None:
        note("bilinear", gridpp.bilinear(grid, pts, field), ref, False, float(np.nanmax(np.abs(field))))
    note("nearest", gridpp.nearest(grid, pts, field), O.nearest(og, op, field), True)
    r = float(rng.uniform(0.02, 0.4) * span)
    note("count", gridpp.count(pts, grid, r).ravel(), O.count(op, og, r), True)
    for stat in ("Mean", "Sum", "Count", "Min", "Max", "Median", "Std", "Variance"):
        mn = int(rng.integers(0, 4))
        note("gridding " + stat, np.asarray(gridpp.gridding(grid, pts, vals, r, mn, getattr(gridpp, stat))).ravel(),
             O.gridding(og, op, vals, r, mn, getattr(refapi, stat)), stat in ("Count", "Min", "Max", "Median"),
             {"Sum": float(np.abs(vals).sum()), "Variance": float(np.max(vals ** 2))}.get(stat, float(np.abs(vals).max())))
    note("gridding_nearest", np.asarray(gridpp.gridding_nearest(grid, pts, vals, 1, gridpp.Mean)).ravel(), O.gridding_nearest(og, op, vals, 1, refapi.Mean), True)
    radii = rng.uniform(0, 0.3 * span, n).astype(np.float32)
    note("fill", gridpp.fill(grid, np.nan_to_num(field), pts, radii, 7.0, bool(seed & 1)), O.fill(og, np.nan_to_num(field), op, radii, 7.0, bool(seed & 1)), True)
    med = float(rng.choice([np.nan, 100.0]))
    note("doping_circle", gridpp.doping_circle(grid, np.nan_to_num(field), pts, vals, radii, med), O.doping_circle(og, np.nan_to_num(field), op, vals, radii, med), True)
    hw = rng.integers(0, 5, n).astype(np.int32)
    note("doping_square", gridpp.doping_square(grid, np.nan_to_num(field), pts, vals, hw, med), O.doping_square(og, (Y, X), np.nan_to_num(field), op, vals, hw, med), True)
    note("fill_missing", gridpp.fill_missing(field), O.fill_missing(field), True)
    search = rng.random((Y, X)).astype(np.float32)
    h = int(rng.integers(0, 4))
    note("neighbourhood_search", gridpp.neighbourhood_search(field, search, h, 0.6, 0.9, 0.1), O.neighbourhood_search(field, search, h, 0.6, 0.9, 0.1), True)
    note("calc_gradient MinMax", gridpp.calc_gradient(elev, field, gridpp.MinMax, h + 1, 2, 20.0, -1.0), O.calc_gradient(elev, field, 0, h + 1, 2, 20.0, -1.0), True)
    k = int(rng.integers(1, 6))
    d_out, d_ref = gridpp.distance(pts, grid, k).ravel(), O.distance(op, og, k, False)
    assert np.all(np.abs(d_out - d_ref) <= np.maximum(1e-5 * d_ref, 0.05)), ("distance", np.abs(d_out - d_ref).max())
print("seeds:", seed)
for k_, v in sorted(worst.items()):
    print("%-24s worst relative deviation %.2e" % (k_, v))
