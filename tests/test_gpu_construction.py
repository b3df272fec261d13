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
