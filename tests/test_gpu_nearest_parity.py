"""Nearest-neighbour indices on the device (brute force and binned index) are bit-exact vs the oracle
(float32 squared chord in the reference's operation order, ties -> lowest index)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,geodetic", [(500, True), (20000, True), (20000, False)])
def test_nn_indices_bit_exact(n, geodetic, monkeypatch):
    import gridpp_amd as gridpp
    from oracle import oracle as O
    rng = np.random.default_rng(n)
    ctype = 0 if geodetic else 1
    scale = 1.0 if geodetic else 100000.0
    lat, lon = rng.random(n) * scale, rng.random(n) * scale
    lat[5], lon[5] = lat[4], lon[4]                      # duplicate point
    qlat = np.concatenate([rng.random(3000) * scale * 1.2 - 0.1 * scale, lat[:50]])   # some outside the hull, some exact matches
    qlon = np.concatenate([rng.random(3000) * scale * 1.2 - 0.1 * scale, lon[:50]])
    pts = gridpp.Points(lat, lon, type=ctype)
    op, oq = O.Pts(lat, lon, ctype=ctype), O.Pts(qlat, qlon, ctype=ctype)
    ref = O.nearest_indices(op, oq)
    got = pts._nearest_flat(qlat, qlon, True)
    np.testing.assert_array_equal(got, ref)
    monkeypatch.setenv("GPP_NN_BRUTE", "1")
    np.testing.assert_array_equal(pts._nearest_flat(qlat, qlon, True), ref)
    monkeypatch.delenv("GPP_NN_BRUTE")
    # include_match = False drops exact coordinate matches (kdtree.cpp:265-270)
    nm = pts._nearest_flat(qlat[-50:], qlon[-50:], False)
    assert (nm != np.arange(50)).all()


def test_nearest_grid_to_points_large():
    import gridpp_amd as gridpp
    from oracle import oracle as O
    Y = X = 300
    lats, lons = np.meshgrid(np.linspace(50, 52, Y), np.linspace(5, 8, X), indexing="ij")
    vals = np.arange(Y * X, dtype=np.float32).reshape(Y, X)
    rng = np.random.default_rng(4)
    plat, plon = 50 + 2 * rng.random(2000), 5 + 3 * rng.random(2000)
    out = gridpp.nearest(gridpp.Grid(lats, lons), gridpp.Points(plat, plon), vals)
    ref = O.nearest(O.Pts(lats.ravel(), lons.ravel()), O.Pts(plat, plon), vals)
    np.testing.assert_array_equal(out, ref)
