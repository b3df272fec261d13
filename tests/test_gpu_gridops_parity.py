"""fill / fill_missing / doping / neighbourhood_search / calc_gradient on the device vs the oracle (C restatements in
oracle/gridpp_oracle.c, pinned on the reference's own tests by tests/test_oracle_golden.py).  Everything here is either a
selection (exact) or float arithmetic in the reference's association (exact); the LinearRegression gradient goes through
box means whose double sums are formed separably instead of from a summed-area table, hence its 1e-5 tolerance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _grid(Y, X, cartesian):
    import gridpp_amd as gridpp
    from oracle import oracle as O
    rng = np.random.default_rng(Y * 1000 + X)
    if cartesian:
        lats, lons = np.meshgrid(np.linspace(0, 50000, Y), np.linspace(0, 70000, X), indexing="ij")
    else:
        lats, lons = np.meshgrid(np.linspace(59, 59.5, Y), np.linspace(10, 11.2, X), indexing="ij")
    elev = rng.uniform(0, 600, (Y, X)).astype(np.float32)
    ct = 1 if cartesian else 0
    return gridpp.Grid(lats, lons, elev, 0 * elev, ct), O.Pts(lats.ravel(), lons.ravel(), elev.ravel(), None, ct), lats, lons, ct


def _points(lats, lons, n, ct, seed):
    import gridpp_amd as gridpp
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    plat = lats.min() + (lats.max() - lats.min()) * (1.2 * rng.random(n) - 0.1)
    plon = lons.min() + (lons.max() - lons.min()) * (1.2 * rng.random(n) - 0.1)
    pelev = rng.uniform(0, 600, n).astype(np.float32)
    return gridpp.Points(plat, plon, pelev, 0 * pelev, ct), O.Pts(plat, plon, pelev, None, ct), rng


@pytest.mark.parametrize("cartesian", [False, True])
def test_fill_matches_oracle(cartesian):
    import gridpp_amd as gridpp
    from oracle import oracle as O
    Y, X = 70, 90
    grid, og, lats, lons, ct = _grid(Y, X, cartesian)
    pts, op, rng = _points(lats, lons, 60, ct, 4)
    radii = rng.uniform(0, 6000, 60).astype(np.float32)
    radii[:3] = 0
    field = rng.normal(0, 1, (Y, X)).astype(np.float32)
    for outside in (False, True):
        out = gridpp.fill(grid, field, pts, radii, -7.5, outside)
        ref = O.fill(og, field, op, radii, -7.5, outside)
        np.testing.assert_array_equal(out, ref)
    assert 0.02 < np.mean(np.asarray(out) == field) < 0.98


@pytest.mark.parametrize("cartesian", [False, True])
@pytest.mark.parametrize("max_elev_diff", [np.nan, 150.0])
def test_doping_matches_oracle(cartesian, max_elev_diff):
    """Overlapping circles / squares: the observation with the highest index wins, as in the reference's sequential loop."""
    import gridpp_amd as gridpp
    from oracle import oracle as O
    Y, X = 60, 75
    grid, og, lats, lons, ct = _grid(Y, X, cartesian)
    n = 120
    pts, op, rng = _points(lats, lons, n, ct, 9)
    obs = rng.normal(0, 5, n).astype(np.float32)
    bg = rng.normal(100, 1, (Y, X)).astype(np.float32)
    radii = rng.uniform(500, 9000, n).astype(np.float32)
    hw = rng.integers(0, 6, n).astype(np.int32)
    out = gridpp.doping_circle(grid, bg, pts, obs, radii, max_elev_diff)
    np.testing.assert_array_equal(out, O.doping_circle(og, bg, op, obs, radii, max_elev_diff))
    assert 0.05 < np.mean(np.asarray(out) != bg) < 0.999
    out = gridpp.doping_square(grid, bg, pts, obs, hw, max_elev_diff)
    np.testing.assert_array_equal(out, O.doping_square(og, (Y, X), bg, op, obs, hw, max_elev_diff))
    with pytest.raises(ValueError):
        gridpp.doping_circle(grid, bg, pts, obs, -radii, max_elev_diff)
    with pytest.raises(ValueError):
        gridpp.doping_square(grid, bg, pts, obs, hw[:-1], max_elev_diff)


def test_fill_missing_matches_oracle(monkeypatch):
    import torch
    import gridpp_amd as gridpp
    from oracle import oracle as O
    rng = np.random.default_rng(6)
    f = rng.normal(0, 3, (83, 121)).astype(np.float32)
    f[rng.random(f.shape) < 0.2] = np.nan
    f[10:30, 40:80] = np.nan          # a hole
    f[:, 0] = np.nan                  # a missing edge column
    f[50, :] = np.nan                 # a missing row
    out = gridpp.fill_missing(f)
    np.testing.assert_array_equal(out, O.fill_missing(f))
    dev = gridpp.fill_missing(torch.from_numpy(f).cuda())
    np.testing.assert_array_equal(dev.cpu().numpy(), out)
    # the one-thread-per-line kernels (fields wider than the LDS row) give the same bits
    monkeypatch.setenv("GPP_FILL_MISSING_LINES", "1")
    np.testing.assert_array_equal(gridpp.fill_missing(f), out)
    monkeypatch.delenv("GPP_FILL_MISSING_LINES")
    # odd shapes: one row, one column, all missing, a missing first element
    for g in (f[:1, :], f[:, :1], np.full((7, 9), np.nan, np.float32), np.array([[np.nan, 1, np.nan, 3, np.nan]], np.float32)):
        np.testing.assert_array_equal(gridpp.fill_missing(np.ascontiguousarray(g)), O.fill_missing(np.ascontiguousarray(g)))


@pytest.mark.parametrize("with_apply", [False, True])
def test_neighbourhood_search_matches_oracle(with_apply):
    import gridpp_amd as gridpp
    from oracle import oracle as O
    rng = np.random.default_rng(16)
    Y, X = 57, 64
    array = rng.normal(10, 4, (Y, X)).astype(np.float32)
    search = rng.random((Y, X)).astype(np.float32)
    array[rng.random((Y, X)) < 0.05] = np.nan
    search[rng.random((Y, X)) < 0.05] = np.nan
    apply = (rng.integers(0, 3, (Y, X))).astype(np.int32) if with_apply else None     # 0: skip, 1: search, 2: neither
    for hw, tmin, tmax, delta in ((1, 0.7, 1.0, 0.1), (3, 0.95, 1.0, 0.3), (2, 2.0, 3.0, 0.05)):
        out = gridpp.neighbourhood_search(array, search, hw, tmin, tmax, delta, apply)
        ref = O.neighbourhood_search(array, search, hw, tmin, tmax, delta, apply)
        np.testing.assert_array_equal(out, ref)
    with pytest.raises(ValueError):
        gridpp.neighbourhood_search(array, search, 1, 1.0, 0.5, 0.1)


@pytest.mark.parametrize("gradient_type", ["MinMax", "LinearRegression"])
def test_calc_gradient_matches_oracle(gradient_type):
    import gridpp_amd as gridpp
    from oracle import oracle as O
    rng = np.random.default_rng(23)
    Y, X = 66, 71
    base = rng.uniform(0, 1500, (Y, X)).astype(np.float32)
    values = (280 - 0.0065 * base + rng.normal(0, 0.3, (Y, X))).astype(np.float32)
    base[20:30, 10:40] = np.nan
    values[rng.random((Y, X)) < 0.03] = np.nan
    gt = getattr(gridpp, gradient_type)
    for hw, num_min, min_range, dflt in ((1, 0, np.nan, -0.0065), (3, 5, 100.0, 0.0), (7, 2, 0.0, 1.0)):
        out = np.asarray(gridpp.calc_gradient(base, values, gt, hw, num_min, min_range, dflt))
        ref = O.calc_gradient(base, values, gt, hw, num_min, min_range, dflt)
        if gradient_type == "MinMax":
            np.testing.assert_array_equal(out, ref)
        else:
            # the gradient divides by a variance formed by cancellation: compare where it is well conditioned, exactly elsewhere on validity
            assert np.array_equal(np.isnan(out), np.isnan(ref))
            np.testing.assert_allclose(out, ref, rtol=2e-3, atol=1e-6)
            assert np.median(np.abs(out - ref) / np.maximum(np.abs(ref), 1e-9)) < 1e-5
    assert (out != dflt).mean() > 0.5


def test_fill_and_doping_properties_at_size():
    """Sizes the oracle cannot reach (2000 x 2000 grid, 50 000 points): inside / outside fills are complements, a doping
    with one value and no elevation test is the inside fill, and counts agree with `count`."""
    import gridpp_amd as gridpp
    N, P = 2000, 50000
    lats, lons = np.meshgrid(np.linspace(58, 60, N), np.linspace(8, 12, N), indexing="ij")
    grid = gridpp.Grid(lats, lons)
    rng = np.random.default_rng(5)
    pts = gridpp.Points(58 + 2 * rng.random(P), 8 + 4 * rng.random(P))
    radii = rng.uniform(100, 1500, P).astype(np.float32)
    field = rng.normal(0, 1, (N, N)).astype(np.float32)          # never equal to the fill value
    inside = np.asarray(gridpp.fill(grid, field, pts, radii, 99.0, False))
    outside = np.asarray(gridpp.fill(grid, field, pts, radii, 99.0, True))
    hit = inside == 99.0
    assert 0.05 < hit.mean() < 0.95
    np.testing.assert_array_equal(outside == 99.0, ~hit)
    np.testing.assert_array_equal(inside[~hit], field[~hit])
    np.testing.assert_array_equal(outside[hit], field[hit])
    doped = np.asarray(gridpp.doping_circle(grid, field, pts, np.full(P, 99.0, np.float32), radii))
    np.testing.assert_array_equal(doped, inside)
    # equal radii: a cell is hit exactly when `count` finds a point within that radius
    same = np.full(P, 800.0, np.float32)
    hit800 = np.asarray(gridpp.fill(grid, field, pts, same, 99.0, False)) == 99.0
    np.testing.assert_array_equal(hit800, gridpp.count(pts, grid, 800.0) > 0)
    # the winner of overlapping circles is the highest index: doping with obs = index
    obs = np.arange(P, dtype=np.float32)
    who = np.asarray(gridpp.doping_circle(grid, np.full((N, N), -1, np.float32), pts, obs, same))
    assert (who >= 0).sum() == hit800.sum() and who.max() == P - 1


def test_device_tensor_inputs_give_the_host_results():
    """torch CUDA tensors in -> CUDA tensors out, same values as the numpy path (fill, doping, neighbourhood_search, calc_gradient)."""
    import torch
    import gridpp_amd as gridpp
    Y, X = 40, 50
    grid, og, lats, lons, ct = _grid(Y, X, False)
    pts, op, rng = _points(lats, lons, 30, ct, 11)
    field = rng.normal(0, 1, (Y, X)).astype(np.float32)
    other = rng.random((Y, X)).astype(np.float32)
    radii = rng.uniform(500, 6000, 30).astype(np.float32)
    obs = rng.normal(0, 2, 30).astype(np.float32)
    d_field, d_other, d_obs = torch.from_numpy(field).cuda(), torch.from_numpy(other).cuda(), torch.from_numpy(obs).cuda()
    pairs = [
        (gridpp.fill(grid, d_field, pts, radii, 3.0, False), gridpp.fill(grid, field, pts, radii, 3.0, False)),
        (gridpp.doping_circle(grid, d_field, pts, d_obs, radii), gridpp.doping_circle(grid, field, pts, obs, radii)),
        (gridpp.doping_square(grid, d_field, pts, d_obs, np.full(30, 2, np.int32)), gridpp.doping_square(grid, field, pts, obs, np.full(30, 2, np.int32))),
        (gridpp.neighbourhood_search(d_field, d_other, 2, 0.7, 1.0, 0.1), gridpp.neighbourhood_search(field, other, 2, 0.7, 1.0, 0.1)),
        (gridpp.calc_gradient(d_other, d_field, gridpp.LinearRegression, 2, 0, 0.0, -1.0), gridpp.calc_gradient(other, field, gridpp.LinearRegression, 2, 0, 0.0, -1.0)),
        (gridpp.calc_gradient(d_other, d_field, gridpp.MinMax, 2, 0, 0.0, -1.0), gridpp.calc_gradient(other, field, gridpp.MinMax, 2, 0, 0.0, -1.0)),
        (gridpp.nearest(grid, pts, d_field), gridpp.nearest(grid, pts, field)),
    ]
    for dev, host in pairs:
        assert dev.is_cuda
        np.testing.assert_array_equal(dev.cpu().numpy(), np.asarray(host))
