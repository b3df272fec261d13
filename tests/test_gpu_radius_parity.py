"""count / gridding / gridding_nearest / get_neighbours on the device vs the oracle (linear scans in
oracle/gridpp_oracle.c, pinned on the reference's tests/test_count.py and tests/test_gridding.py).  Counts, indices, Count /
Min / Max / Median are exact; Mean / Sum / Std / Variance accumulate in float32 in a different neighbour order than the
oracle's (the reference's own order is the R-tree's, i.e. unspecified), hence the 1e-5 relative tolerance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def _sets(n_in, n_out, geodetic, seed):
    rng = np.random.default_rng(seed)
    scale = 1.0 if geodetic else 100000.0
    ilat, ilon = rng.random(n_in) * scale, rng.random(n_in) * scale
    olat, olon = rng.random(n_out) * scale * 1.2 - 0.1 * scale, rng.random(n_out) * scale * 1.2 - 0.1 * scale
    olat[:20], olon[:20] = ilat[:20], ilon[:20]          # exact matches
    return ilat, ilon, olat, olon


@pytest.mark.parametrize("n_in,geodetic", [(300, True), (5000, True), (5000, False), (200000, True)])
def test_count_is_exact(n_in, geodetic):
    import gridpp_amd as gridpp
    from oracle import oracle as O
    ct = 0 if geodetic else 1
    n_out = 1500 if n_in <= 5000 else 300
    ilat, ilon, olat, olon = _sets(n_in, n_out, geodetic, n_in)
    ip, op = gridpp.Points(ilat, ilon, type=ct), gridpp.Points(olat, olon, type=ct)
    oi_, oo = O.Pts(ilat, ilon, ctype=ct), O.Pts(olat, olon, ctype=ct)
    for radius in (0.0, 800.0, 7000.0, 3e5):
        np.testing.assert_array_equal(gridpp.count(ip, op, radius), O.count(oi_, oo, radius))
    assert gridpp.count(ip, op, 7000.0).max() > 3


def test_count_grid_overloads_and_shapes():
    import gridpp_amd as gridpp
    from oracle import oracle as O
    lats, lons = np.meshgrid(np.linspace(59, 60, 40), np.linspace(10, 12, 50), indexing="ij")
    rng = np.random.default_rng(1)
    plat, plon = 59 + rng.random(700), 10 + 2 * rng.random(700)
    grid, pts = gridpp.Grid(lats, lons), gridpp.Points(plat, plon)
    og, op = O.Pts(lats.ravel(), lons.ravel()), O.Pts(plat, plon)
    out = gridpp.count(pts, grid, 5000)
    assert out.shape == (40, 50)
    np.testing.assert_array_equal(out.ravel(), O.count(op, og, 5000))
    np.testing.assert_array_equal(gridpp.count(grid, pts, 5000), O.count(og, op, 5000))
    np.testing.assert_array_equal(gridpp.count(grid, grid, 3000).ravel(), O.count(og, og, 3000))


@pytest.mark.parametrize("statistic", ["Mean", "Sum", "Count", "Min", "Max", "Median", "Std", "Variance"])
def test_gridding_matches_oracle(statistic, monkeypatch):
    import gridpp_amd as gridpp
    from oracle import oracle as O
    from tests import refapi
    lats, lons = np.meshgrid(np.linspace(59, 60, 30), np.linspace(10, 12, 45), indexing="ij")
    rng = np.random.default_rng(8)
    n = 4000
    plat, plon = 58.9 + 1.2 * rng.random(n), 9.9 + 2.2 * rng.random(n)
    values = rng.normal(5, 3, n).astype(np.float32)
    values[rng.random(n) < 0.02] = np.nan
    grid, pts = gridpp.Grid(lats, lons), gridpp.Points(plat, plon)
    og, op = O.Pts(lats.ravel(), lons.ravel()), O.Pts(plat, plon)
    stat = getattr(gridpp, statistic)
    for radius, min_num in ((4000.0, 0), (9000.0, 3), (2500.0, 7)):
        out = np.asarray(gridpp.gridding(grid, pts, values, radius, min_num, stat))
        ref = O.gridding(og, op, values, radius, min_num, getattr(refapi, statistic)).reshape(30, 45)
        assert np.array_equal(np.isnan(out), np.isnan(ref))
        if statistic in ("Count", "Min", "Max", "Median"):
            np.testing.assert_array_equal(out, ref)
        else:
            np.testing.assert_allclose(out, ref, rtol=RTOL, atol=1e-5)
    assert np.isfinite(out).any() and np.isnan(out).any()
    # the one-pass kernel and the CSR path (used for Median) feed the accumulators in the same order
    for path in ("GPP_GRIDDING_CSR", "GPP_GRIDDING_THREAD"):      # default = one wavefront per location
        monkeypatch.setenv(path, "1")
        np.testing.assert_array_equal(np.asarray(gridpp.gridding(grid, pts, values, 2500.0, 7, stat)), out)
        monkeypatch.delenv(path)
    # Points as the output set and device-resident values
    import torch
    outp = gridpp.gridding(grid.to_points(), pts, torch.from_numpy(values).cuda(), 9000.0, 3, stat)
    assert outp.is_cuda
    np.testing.assert_allclose(outp.cpu().numpy(), np.asarray(gridpp.gridding(grid, pts, values, 9000.0, 3, stat)).ravel(), rtol=0, atol=0,
                               equal_nan=True)


@pytest.mark.parametrize("statistic", ["Mean", "Sum", "Count", "Max", "Median", "Std"])
def test_gridding_nearest_is_exact(statistic):
    """The values of a location arrive in input order on both sides, so every statistic is bit-exact."""
    import gridpp_amd as gridpp
    from oracle import oracle as O
    from tests import refapi
    lats, lons = np.meshgrid(np.linspace(59, 60, 25), np.linspace(10, 12, 35), indexing="ij")
    rng = np.random.default_rng(3)
    n = 6000
    plat, plon = 58.9 + 1.2 * rng.random(n), 9.9 + 2.2 * rng.random(n)
    values = rng.normal(0, 4, n).astype(np.float32)
    values[::97] = np.nan
    grid, pts = gridpp.Grid(lats, lons), gridpp.Points(plat, plon)
    og, op = O.Pts(lats.ravel(), lons.ravel()), O.Pts(plat, plon)
    for min_num in (0, 1, 8):
        out = np.asarray(gridpp.gridding_nearest(grid, pts, values, min_num, getattr(gridpp, statistic)))
        ref = O.gridding_nearest(og, op, values, min_num, getattr(refapi, statistic)).reshape(25, 35)
        np.testing.assert_array_equal(out, ref)
    out = np.asarray(gridpp.gridding_nearest(grid.to_points(), pts, values, 0, getattr(gridpp, statistic)))
    np.testing.assert_array_equal(out, O.gridding_nearest(og, op, values, 0, getattr(refapi, statistic)))


def test_get_neighbours_with_distance_and_closest():
    import gridpp_amd as gridpp
    from oracle import oracle as O
    rng = np.random.default_rng(12)
    n = 30000
    lat, lon = 60 + rng.random(n), 10 + rng.random(n)
    lat[7], lon[7] = lat[3], lon[3]
    pts = gridpp.Points(lat, lon)
    op = O.Pts(lat, lon)
    for qlat, qlon, r in ((60.5, 10.5, 3000.0), (float(np.float32(lat[3])), float(np.float32(lon[3])), 1500.0), (61.5, 12.0, 5000.0)):
        for inc in (True, False):
            ref = O.get_neighbours(op, qlat, qlon, r, inc)
            np.testing.assert_array_equal(pts.get_neighbours(qlat, qlon, r, inc), ref)
            assert pts.get_num_neighbours(qlat, qlon, r, inc) == len(ref)
            idx, dist = pts.get_neighbours_with_distance(qlat, qlon, r, inc)
            np.testing.assert_array_equal(idx, ref)
            assert len(dist) == len(ref) and (len(ref) == 0 or (np.asarray(dist) <= r).all())
    # k nearest: distance-sorted, ties -> lower index; 3 and 7 coincide
    q = (float(np.float32(lat[3])), float(np.float32(lon[3])))
    assert list(pts.get_closest_neighbours(q[0], q[1], 2)) == [3, 7]
    got = pts.get_closest_neighbours(60.25, 10.75, 50)
    x, y, z = O.convert_coordinates([60.25], [10.75])
    d2 = (op.x - x[0]) ** 2 + (op.y - y[0]) ** 2 + (op.z - z[0]) ** 2
    assert len(got) == 50 and set(got) == set(np.argsort(d2, kind="stable")[:50])
    assert 3 not in pts.get_closest_neighbours(q[0], q[1], 3, False) and 7 not in pts.get_closest_neighbours(q[0], q[1], 3, False)


def test_reference_benchmark_shape_gridding():
    """tests/benchmark.py:61-62: 200 x 200 grid, 100 000 points on the diagonal, radius 5 km, min_num 1, Mean."""
    import gridpp_amd as gridpp
    y, x = np.meshgrid(np.linspace(0, 1, 200), np.linspace(0, 1, 200))
    grid = gridpp.Grid(y, x, 0 * x, 0 * x)
    n = 100000
    pts = gridpp.Points(np.linspace(0, 1, n), np.linspace(0, 1, n), np.zeros(n), np.zeros(n))
    values = np.arange(n, dtype=np.float32) / n
    cnt = gridpp.count(pts, grid, 5000)
    out = np.asarray(gridpp.gridding(grid, pts, values, 5000, 1, gridpp.Mean))
    assert np.array_equal(np.isnan(out), cnt == 0)
    # the points within 5 km of a grid node on the diagonal are symmetric around it: their mean is the node's own coordinate
    k = np.arange(20, 180)
    np.testing.assert_allclose(out[k, k], k / 199.0, atol=2e-3)
    near = np.asarray(gridpp.gridding_nearest(grid, pts, values, 1, gridpp.Count))
    assert np.nansum(near) == n


@pytest.mark.parametrize("geodetic", [True, False])
def test_distance_matches_oracle(geodetic):
    """distance.cpp: the k nearest set is exact (same float32 metric and tie rule); the great-circle arc is double trigonometry
    on float32-rounded radians, where device and host libm may differ in the last bits of cos / sin / acos: 1e-5 relative or
    5 cm, whichever is larger (the planar form is exact)."""
    import gridpp_amd as gridpp
    from oracle import oracle as O
    ct = 0 if geodetic else 1
    ilat, ilon, olat, olon = _sets(2500, 400, geodetic, 77)
    ip, op = gridpp.Points(ilat, ilon, type=ct), gridpp.Points(olat, olon, type=ct)
    oi_, oo = O.Pts(ilat, ilon, ctype=ct), O.Pts(olat, olon, ctype=ct)
    lats, lons = np.meshgrid(np.linspace(olat.min(), olat.max(), 20), np.linspace(olon.min(), olon.max(), 25), indexing="ij")
    ogrid, oog = gridpp.Grid(lats, lons, type=ct), O.Pts(lats.ravel(), lons.ravel(), ctype=ct)
    for num in (1, 3, 10, 200, 3000):            # the last one asks for more points than there are
        out = gridpp.distance(ip, op, num)
        ref = O.distance(oi_, oo, num, True)
        if geodetic:
            np.testing.assert_allclose(out, ref, rtol=1e-5, atol=0.05)
        else:
            np.testing.assert_array_equal(out, ref)
        outg = gridpp.distance(ip, ogrid, num)
        refg = O.distance(oi_, oog, num, False).reshape(20, 25)
        assert outg.shape == (20, 25)
        if geodetic:
            np.testing.assert_allclose(outg, refg, rtol=1e-5, atol=0.05)
        else:
            np.testing.assert_array_equal(outg, refg)
    assert (gridpp.distance(ip, op, 1)[:20] == 0).all()          # exact matches
    with pytest.raises(ValueError):
        gridpp.distance(ip, gridpp.Points(olat, olon, type=1 - ct), 1)


def test_gridding_median_in_chunks(monkeypatch):
    """The CSR path splits the locations into passes when the neighbour lists exceed the scratch cap; with a tiny cap the
    passes must reproduce the single-pass result (a location with more neighbours than the cap gets a pass of its own)."""
    import gridpp_amd as gridpp
    rng = np.random.default_rng(31)
    n = 3000
    plat, plon = 59 + rng.random(n), 10 + 2 * rng.random(n)
    values = rng.normal(0, 2, n).astype(np.float32)
    lats, lons = np.meshgrid(np.linspace(59, 60, 23), np.linspace(10, 12, 31), indexing="ij")
    grid, pts = gridpp.Grid(lats, lons), gridpp.Points(plat, plon)
    one = np.asarray(gridpp.gridding(grid, pts, values, 12000.0, 2, gridpp.Median))
    cnt = gridpp.count(pts, grid, 12000.0)
    for cap in (int(cnt.max()) // 2, 1000, 20000):
        monkeypatch.setenv("GPP_CSR_CAP", str(cap))
        np.testing.assert_array_equal(np.asarray(gridpp.gridding(grid, pts, values, 12000.0, 2, gridpp.Median)), one)
    monkeypatch.delenv("GPP_CSR_CAP")
    assert cnt.sum() > 20000
