"""Host-side helpers of the python mirror that need no GPU: the KDTree static distance functions, pinned with the
reference's own known-answer values (tests/test_kdtree.py:54-147)."""
import numpy as np

import gridpp_amd as gridpp


def test_rad2deg():
    assert abs(gridpp.KDTree_rad2deg(1) - 180 / 3.14159265) < 1e-5
    assert abs(gridpp.KDTree_rad2deg(-1) + 180 / 3.14159265) < 1e-5
    assert gridpp.KDTree_rad2deg(0) == 0


def test_calc_distance():
    cfg = [[60, 10, 61, 11, 0.1, 124080.79], [60, 10, 60, 10, 0, 0], [90, 10, -90, 10, 0, 20037508],
           [0, 0, 0, 180, 0, 20037508], [60.5, 5.25, -84.75, -101.75, 0, 16879114]]
    for lat0, lon0, lat1, lon1, delta, expected in cfg:
        p0, p1 = gridpp.Point(lat0, lon0), gridpp.Point(lat1, lon1)
        for d in (gridpp.KDTree.calc_distance(p0, p1), gridpp.KDTree.calc_distance(lat0, lon0, lat1, lon1)):
            assert abs(d - expected) <= max(delta, 1e-7 * expected + 1e-6), (lat0, lon0, lat1, lon1, d)


def test_calc_straight_distance_and_limit():
    cfg = [[60, 10, 61, 11, 2, 124080.79], [60, 10, 60, 10, 0, 0], [90, 10, -90, 10, 0, 6.378137e6 * 2],
           [0, 0, 0, 180, 10, 6.378137e6 * 2], [60.5, 5.25, -84.75, -101.75, 0, 12367265.0]]
    for lat0, lon0, lat1, lon1, delta, expected in cfg:
        p0, p1 = gridpp.Point(lat0, lon0), gridpp.Point(lat1, lon1)
        for d in (gridpp.KDTree.calc_straight_distance(p0, p1),
                  gridpp.KDTree.calc_straight_distance(p0.x, p0.y, p0.z, p1.x, p1.y, p1.z)):
            assert abs(d - expected) <= max(delta, 1e-7 * expected + 1e-6)
    p0, p1 = gridpp.Point(0, 0), gridpp.Point(0.001, 0.001)
    assert abs(gridpp.KDTree_calc_distance(0, 0, 0.001, 0.001) - 157.42953491210938) < 1e-7 * 157
    assert abs(gridpp.KDTree_calc_straight_distance(p0.x, p0.y, p0.z, p1.x, p1.y, p1.z) - 157.42953491210938) < 1e-7 * 157


def test_calc_distance_fast():
    cfg = [[60, 10, 60, 10, 10, 0], [90, 10, -90, 10, 10, 20037508], [0, 0, 0, 180, 10, 20037508], [60, 10, 61, 11, 400, 124080.79],
           [89, 0, 90, 0, 10, 111319.62], [89, 0, 90, 180, 10, 111319.62], [89, 0, 89.9, 180, 6000, 111319.62],
           [0, 179, 0, -179, 100, 222639.64]]
    for lat in (-90, -89, 0, 89, 90):
        cfg.append([lat, 180, lat, -180, 10, 0])
    for lat in (-90, -89, 0, 89):
        cfg.append([lat, 180, lat + 1, -180, 10, 111319.4921875])
    for lat0, lon0, lat1, lon1, delta, expected in cfg:
        assert abs(gridpp.KDTree.calc_distance_fast(lat0, lon0, lat1, lon1) - expected) <= delta


def test_point_coordinates():
    p = gridpp.Point(60, 10)
    x, y, z = gridpp.convert_coordinates([60], [10])
    assert (p.x, p.y, p.z) == (float(x[0]), float(y[0]), float(z[0]))
    q = gridpp.Point(3, 4, 0, 0, gridpp.Cartesian)   # src/api/point.cpp:18-21: x = lat, y = lon
    assert (q.x, q.y, q.z) == (3.0, 4.0, 0.0)
