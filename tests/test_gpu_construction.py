"""Large point sets are converted and indexed on the device: the coordinates must equal the host (libm) conversion bit
for bit, and nearest-neighbour lookups through the device-built index must equal the host-built index."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_device_conversion_is_bit_identical_to_libm():
    import gridpp_amd as gridpp
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    n = 3_000_000
    lats = np.concatenate([rng.uniform(-90, 90, n - 8), [0, 90, -90, 45, 1e-30, 89.99999, -0.0, 60]]).astype(np.float32)
    lons = np.concatenate([rng.uniform(-360, 360, n - 8), [0, 90, 180, 270, 1e-30, 359.99997, -180, 10]]).astype(np.float32)
    p = gridpp.Points(lats, lons)                      # n >= 65536: device conversion
    x, y, z = O.convert_coordinates(lats, lons, 0)     # host libm (the reference's arithmetic)
    gx, gy, gz = p._field(4), p._field(5), p._field(6)   # x, y, z as held by the library
    np.testing.assert_array_equal(gx, x)
    np.testing.assert_array_equal(gy, y)
    np.testing.assert_array_equal(gz, z)
    # cartesian: pass-through
    pc = gridpp.Points(lats, lons, (), (), gridpp.Cartesian)
    np.testing.assert_array_equal(pc._field(4), lons)
    np.testing.assert_array_equal(pc._field(5), lats)


def test_invalid_coordinates_raise_on_the_device_path():
    import gridpp_amd as gridpp
    lats = np.zeros(100000, np.float32)
    lons = np.zeros(100000, np.float32)
    lats[77777] = 91.0
    with pytest.raises(Exception):
        gridpp.Points(lats, lons)


def test_device_built_index_gives_the_same_nearest_neighbours():
    import gridpp_amd as gridpp
    rng = np.random.default_rng(6)
    ny = nx = 600                                       # 360 000 points: device index
    lats, lons = np.meshgrid(np.linspace(50, 60, ny), np.linspace(0, 20, nx), indexing="ij")
    lats = lats + rng.normal(0, 0.003, lats.shape)      # jittered so that ties are rare but bins are uneven
    values = rng.normal(0, 1, (ny, nx)).astype(np.float32)
    q = gridpp.Points(rng.uniform(49.5, 60.5, 20000), rng.uniform(-0.5, 20.5, 20000))
    a = gridpp.nearest(gridpp.Grid(lats, lons), q, values)
    os.environ["GPP_HOST_INDEX"] = "1"
    os.environ["GPP_HOST_CONVERT"] = "1"
    try:
        b = gridpp.nearest(gridpp.Grid(lats, lons), q, values)
    finally:
        del os.environ["GPP_HOST_INDEX"], os.environ["GPP_HOST_CONVERT"]
    np.testing.assert_array_equal(a, b)
    # and against brute force on a sample
    from oracle import oracle as O
    idx = O.nearest_indices(O.Pts(lats.ravel(), lons.ravel()), O.Pts(q.get_lats()[:300], q.get_lons()[:300]))
    np.testing.assert_array_equal(a[:300], values.ravel()[idx])


def test_float64_inputs_are_cast_on_the_device_like_astype():
    """Grid / Points from float64 arrays (numpy's default) take the f64 entry points above 65 536 points: the device cast must
    give the float32 values numpy's astype gives, and from there the same coordinates as the float32 path."""
    import gridpp_amd as gridpp
    rng = np.random.default_rng(41)
    Y, X = 300, 400
    lats = 50 + 10 * rng.random((Y, X))
    lons = -20 + 60 * rng.random((Y, X))
    elevs = 2000 * rng.random((Y, X))
    a = gridpp.Grid(lats, lons, elevs)                                   # float64 -> f64 entry point
    b = gridpp.Grid(lats.astype(np.float32), lons.astype(np.float32), elevs.astype(np.float32))
    for k in range(7):
        np.testing.assert_array_equal(a._field(k), b._field(k))
    np.testing.assert_array_equal(a.get_lafs(), b.get_lafs())            # absent: NaN on both
    n = 100000
    plat, plon = 50 + 10 * rng.random(n), -20 + 60 * rng.random(n)
    pa, pb = gridpp.Points(plat, plon), gridpp.Points(plat.astype(np.float32), plon.astype(np.float32))
    for k in (0, 1, 4, 5, 6):
        np.testing.assert_array_equal(pa._field(k), pb._field(k))
    q = (55.0, 3.0)
    assert pa.get_nearest_neighbour(*q) == pb.get_nearest_neighbour(*q)
    with pytest.raises(ValueError):
        gridpp.Points(np.full(n, 95.0), plon)                            # invalid latitude is still reported


def test_float64_field_inputs_equal_the_float32_path():
    """GPP_HOST_F64: large float64 numpy fields are cast on the device; results must equal those of the same call on
    astype(float32) inputs bit for bit (OI, neighbourhood 2-D / 3-D, nearest, bilinear)."""
    import gridpp_amd as gridpp
    rng = np.random.default_rng(43)
    Y = X = 1100                                         # > 2^20 cells: the float64 route
    lats, lons = np.meshgrid(np.linspace(59, 60, Y), np.linspace(10, 12, X), indexing="ij")
    grid = gridpp.Grid(lats, lons)
    S = 400
    pts = gridpp.Points(59 + rng.random(S), 10 + 2 * rng.random(S))
    bg = rng.normal(0, 1, (Y, X))                        # float64
    obs, ratios, pbg = rng.normal(0, 1, S), rng.uniform(0.1, 1, S), rng.normal(0, 1, S)
    st = gridpp.BarnesStructure(8000)
    a = gridpp.optimal_interpolation(grid, bg, pts, obs, ratios, pbg, st, 10)
    b = gridpp.optimal_interpolation(grid, bg.astype(np.float32), pts, obs.astype(np.float32), ratios.astype(np.float32),
                                     pbg.astype(np.float32), st, 10)
    np.testing.assert_array_equal(a, b)
    assert np.abs(np.asarray(a) - bg).max() > 0.1
    bg[5:9, 7:300] = np.nan
    np.testing.assert_array_equal(gridpp.neighbourhood(bg, 3, gridpp.Mean), gridpp.neighbourhood(bg.astype(np.float32), 3, gridpp.Mean))
    cube = rng.normal(0, 1, (300, 400, 10))
    np.testing.assert_array_equal(gridpp.neighbourhood(cube, 2, gridpp.Max), gridpp.neighbourhood(cube.astype(np.float32), 2, gridpp.Max))
    np.testing.assert_array_equal(gridpp.nearest(grid, pts, bg), gridpp.nearest(grid, pts, bg.astype(np.float32)))
    np.testing.assert_array_equal(gridpp.bilinear(grid, pts, bg), gridpp.bilinear(grid, pts, bg.astype(np.float32)))
    thr = np.linspace(-2, 2, 7)
    np.testing.assert_array_equal(gridpp.neighbourhood_quantile_fast(cube, 0.5, 2, thr), gridpp.neighbourhood_quantile_fast(cube.astype(np.float32), 0.5, 2, thr))
    # EnSI: (Y, X, E) background in float64
    Ye, Xe, E = 160, 180, 40                             # 1.15 M values
    la, lo = np.meshgrid(np.linspace(59, 59.3, Ye), np.linspace(10, 10.5, Xe), indexing="ij")
    g2 = gridpp.Grid(la, lo)
    p2 = gridpp.Points(59 + 0.3 * rng.random(60), 10 + 0.5 * rng.random(60))
    bge = rng.normal(0, 1, (Ye, Xe, E))
    pbe, ob, sg = rng.normal(0, 1, (60, E)), rng.normal(0, 1, 60), rng.uniform(0.5, 1, 60)
    ea = gridpp.optimal_interpolation_ensi(g2, bge, p2, ob, sg, pbe, st, 8)
    eb = gridpp.optimal_interpolation_ensi(g2, bge.astype(np.float32), p2, ob.astype(np.float32), sg.astype(np.float32), pbe.astype(np.float32), st, 8)
    np.testing.assert_array_equal(ea, eb)
    # the 2-D grid operations
    base = rng.uniform(0, 1500, (Y, X))
    vals = 280 - 0.0065 * base + rng.normal(0, 0.3, (Y, X))
    vals[10:20, 30:90] = np.nan
    for gt in (gridpp.MinMax, gridpp.LinearRegression):
        np.testing.assert_array_equal(gridpp.calc_gradient(base, vals, gt, 3, 2, 10.0, -1.0),
                                      gridpp.calc_gradient(base.astype(np.float32), vals.astype(np.float32), gt, 3, 2, 10.0, -1.0))
    sr = rng.random((Y, X))
    np.testing.assert_array_equal(gridpp.neighbourhood_search(vals, sr, 2, 0.7, 1.0, 0.1),
                                  gridpp.neighbourhood_search(vals.astype(np.float32), sr.astype(np.float32), 2, 0.7, 1.0, 0.1))
    np.testing.assert_array_equal(gridpp.fill_missing(vals), gridpp.fill_missing(vals.astype(np.float32)))
