"""The CPU baseline that bench.py times (oracle.oi_baseline: cell-list radius query + OpenMP, orc_oi_full_omp) must return the
very bits of the serial linear-scan oracle that the parity tests use -- it is the same arithmetic on the same candidates."""
import numpy as np
import pytest

from oracle import oracle as O


@pytest.mark.parametrize("seed,mp,elev", [(1, 20, False), (2, 5, True), (3, 0, False)])
def test_baseline_equals_serial_oracle(seed, mp, elev):
    rng = np.random.default_rng(seed)
    Y = X = 48
    S = 400
    lats, lons = np.meshgrid(np.linspace(0, 1, Y), np.linspace(0, 1, X), indexing="ij")
    bg = rng.normal(0, 1, Y * X).astype(np.float32)
    bg[::97] = np.nan
    plat, plon = rng.random(S), rng.random(S)
    ge = rng.uniform(0, 800, Y * X) if elev else None
    pe = rng.uniform(0, 800, S) if elev else None
    g, p = O.Pts(lats.ravel(), lons.ravel(), ge), O.Pts(plat, plon, pe)
    obs = rng.normal(0, 1, S).astype(np.float32)
    obs[::31] = np.nan
    ratios = rng.uniform(0.1, 1, S).astype(np.float32)
    pbg = rng.normal(0, 1, S).astype(np.float32)
    st = O.Barnes(8000 if mp else 4000, 300 if elev else 0)
    ref = O.oi(g, bg, p, obs, ratios, pbg, st, mp)
    for threads, cell_list in ((1, True), (4, True), (3, False)):
        out = O.oi_baseline(g, bg, p, obs, ratios, pbg, st, mp, threads=threads, cell_list=cell_list)
        assert np.array_equal(out, ref, equal_nan=True), (threads, cell_list)
    assert np.nanmax(np.abs(ref - bg)) > 0.05
