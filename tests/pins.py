"""Known-answer checks of the reference's own unit tests, restated against an
API object `g` that looks like the reference's python module (tests/refapi.py
for the oracle, gridpp_amd for the HIP path).  Expected values come from
tests/golden/reference_known_answers.json (each entry cites the reference test
file:line).  Every function here is called by test_oracle_golden.py (CPU) and
test_gpu_reference_pins.py (GPU)."""
import numpy as np


def _values5(G):
    values = np.reshape(np.arange(25), [5, 5]).astype(np.float32)
    for i, j in G["neighbourhood_values"]["nan_at"]:
        values[i, j] = np.nan
    return values


def pin_barnes_basic(g, G):
    e = G["barnes_basic"]
    s = g.BarnesStructure(e["h"])
    for x, c in zip(e["x"], e["corr"]):
        p1 = g.Point(0, 0, 0, 0, g.Cartesian)
        p2 = g.Point(x, 0, 0, 0, g.Cartesian)
        for f in (s.corr, s.corr_background):
            assert f(p1, p2) == np.float32(c), (x, f(p1, p2), c)
            assert f(p2, p1) == np.float32(c)
            if not np.isnan(x):
                assert f(p2, p2) == 1


def pin_barnes_hmax(g, G):
    e = G["barnes_hmax"]
    p0 = g.Point(0, 0, 0, 0, g.Cartesian)
    for hmax in e["hmaxs"]:
        for dist, ans in e["dist_ans"].items():
            s = g.BarnesStructure(e["h"], 0, 0, hmax)
            corr = s.corr(p0, g.Point(float(dist), 0, 0, 0, g.Cartesian))
            if float(dist) > hmax:
                assert corr == 0
            else:
                assert corr == np.float32(ans), (hmax, dist, corr)


def pin_barnes_invalid(g, G):
    import pytest
    for h in (-1, np.nan):
        with pytest.raises(Exception):
            g.BarnesStructure(h)
    with pytest.raises(Exception):
        g.BarnesStructure(2000, 100, 0, -1)


def _oi_1d_setup(g, G):
    e = G["oi_simple_1d"]
    y, x = e["grid_y"], e["grid_x"]
    grid = g.Grid(y, x, y, y, g.Cartesian)
    points = g.Points(e["points_y"], e["points_x"], [0], [0], g.Cartesian)
    structure = g.BarnesStructure(e["h"])
    return e, grid, points, structure


def pin_oi_simple_1d(g, G):
    e, grid, points, structure = _oi_1d_setup(g, G)
    background = np.zeros([1, 3])
    out = g.optimal_interpolation(grid, background, points, e["pobs"], e["pratios"], e["pbackground"], structure, e["max_points"])
    out = np.asarray(out)
    assert out.dtype == np.float32 and out.shape == (1, 3)
    np.testing.assert_array_almost_equal(out, np.array([[np.exp(-0.5) / 1.1, 1 / 1.1, np.exp(-0.5 * 9) / 1.1]]), e["decimals"])
    np.testing.assert_array_almost_equal(out, np.array(e["expected"]), e["decimals"])


def pin_oi_variance(g, G):
    e, grid, points, structure = _oi_1d_setup(g, G)
    v = G["oi_variance"]
    out, var = g.optimal_interpolation_full(grid, np.zeros([1, 3]), np.ones([1, 3]), points, e["pobs"], [0.1], [0], [1],
                                            structure, e["max_points"])
    assert abs(np.asarray(var)[0, 1] - v["expected_variance_at_obs"]) < 10 ** -v["places"]
    # points overload (tests/test_optimal_interpolation.py:86-105)
    gp = g.Points([0, 0, 0], [0, 2500, 10000], [0, 0, 0], [0, 0, 0], g.Cartesian)
    out, var = g.optimal_interpolation_full(gp, np.zeros(3), np.ones(3), points, np.array([1]), np.array([0.1]),
                                            np.array([0]), np.array([1]), structure, e["max_points"])
    assert abs(np.asarray(var)[1] - v["expected_variance_at_obs"]) < 10 ** -v["places"]


def pin_oi_invalid_arguments(g, G):
    import pytest
    ok = dict(grid=g.Grid([[0, 0, 0]], [[0, 2500, 10000]], [[0, 0, 0]], [[0, 0, 0]], g.Cartesian),
              background=np.zeros([1, 3]),
              points=g.Points([0], [2500], [0], [0], g.Cartesian),
              pobs=[1], pratios=[0.1], pbackground=[0], structure=g.BarnesStructure(2500), max_points=10)
    x = np.zeros([3, 2])
    invalid = {
        'grid': [g.Grid(x, x, x, x, g.Cartesian), g.Grid([[0, 0, 0]], [[0, 2500, 10000]])],
        'points': [g.Points([0, 1], [0, 2500], [0, 0], [0, 0], g.Cartesian), g.Points([0], [2500])],
        'pratios': [np.zeros(11)], 'pobs': [np.zeros([11])], 'background': [np.zeros([2, 11])],
        'pbackground': [np.zeros(21)], 'max_points': [-1]}
    for key, args in invalid.items():
        for arg in args:
            a = dict(ok)
            a[key] = arg
            with pytest.raises(ValueError):
                g.optimal_interpolation(a['grid'], a['background'], a['points'], a['pobs'], a['pratios'],
                                        a['pbackground'], a['structure'], a['max_points'])


def pin_oi_missing_values(g, G):
    e = G["oi_missing_values"]
    obs = np.array(e["obs"], np.float32)
    N = len(obs)
    y = np.arange(0, N * 1000, 1000)
    background = np.zeros(N)
    z = np.zeros(N)
    points = g.Points(y, z, z, z, g.Cartesian)
    ratios = np.ones(N)
    structure = g.BarnesStructure(e["h"], e["v"])
    analysis = g.optimal_interpolation(points, background, points, obs, ratios, background, structure, e["max_points"])
    # the reference test keeps all points (its index set I is "y is not NaN"); the stronger statement -- NaN obs
    # are ignored -- is what the test title says and what src/api/oi.cpp:252 does:
    I = np.where(~np.isnan(obs))[0]
    zI = np.zeros(len(I))
    points1 = g.Points(y[I], zI, zI, zI, g.Cartesian)
    analysis1 = g.optimal_interpolation(points, background, points1, obs[I], ratios[I], background[I], structure, e["max_points"])
    np.testing.assert_array_almost_equal(analysis, analysis1)
    assert not np.isnan(np.asarray(analysis)).any()


def pin_oi_extrapolation(g, G):
    e = G["oi_extrapolation"]
    a, b, n = e["grid_y_linspace"]
    y = np.linspace(a, b, n)
    x = np.zeros(n)
    grid = g.Points(y, x, x, x, g.Cartesian)
    py = e["points_y"]
    z4 = [0] * len(py)
    points = g.Points(py, z4, z4, z4, g.Cartesian)
    pratios = e["pratio"] * np.ones(len(py))
    structure = g.BarnesStructure(e["h"])
    background = np.zeros(n)
    pbackground = np.zeros(len(py))
    o0 = np.asarray(g.optimal_interpolation(grid, background, points, e["pobs"], pratios, pbackground, structure, e["max_points"], False))
    o1 = np.asarray(g.optimal_interpolation(grid, background, points, e["pobs"], pratios, pbackground, structure, e["max_points"], True))
    assert np.max(o0) == 1
    assert np.max(o1) > 1
    I = np.where(o1 < 1)[0]
    np.testing.assert_array_almost_equal(o0[I], o1[I])


def pin_oi_no_obs(g, G):
    grid = g.Points([0], [0])
    points = g.Points([], [])
    out = g.optimal_interpolation(grid, np.zeros(1), points, [], [], [], g.BarnesStructure(500), 10)
    np.testing.assert_almost_equal(out, np.zeros(1))


def pin_ensi(g, G):
    # tests/test_optimal_interpolation_ens.py:9-35
    grid = g.Points([0], [0])
    E = 3
    structure = g.BarnesStructure(500000)
    background = np.zeros([1, E])
    out = g.optimal_interpolation_ensi(grid, background, g.Points([], []), [], [], np.zeros([0, E]), structure, 10)
    np.testing.assert_almost_equal(out, background)
    points = g.Points([0, 0.1], [0, 0.1])
    out = g.optimal_interpolation_ensi(grid, background, points, [np.nan, 0], [1, 1], np.zeros([2, E]), structure, 10)
    np.testing.assert_almost_equal(out, background)


def pin_radius_queries(g, G):
    for key in ("radius_match", "points_neighbours"):
        e = G[key]
        z = [0] * len(e["points_y"])
        points = g.Points(e["points_y"], e["points_x"], z, z, g.Cartesian)
        for q in e["queries"]:
            lat, lon, r, inc = q["q"]
            np.testing.assert_array_equal(np.sort(points.get_neighbours(lat, lon, r, inc)), q["expected"])
        for q in e.get("nearest", []):
            assert points.get_nearest_neighbour(*q["q"]) == q["expected"]
    for key in ("kdtree_geodetic", "kdtree_duplicates", "kdtree_pole"):
        e = G[key]
        tree = g.KDTree(e["lats"], e["lons"])
        for q in e["queries"]:
            lat, lon, r, inc = q["q"]
            np.testing.assert_array_equal(np.sort(tree.get_neighbours(lat, lon, r, inc)), q["expected"])
    e = G["points_nearest_no_match"]
    points = g.Points(e["lats"], e["lons"])
    for c in e["cases"]:
        lat, lon, inc = c["q"]
        assert points.get_nearest_neighbour(lat, lon, inc) in c["expected"]


def pin_invalid_coords(g, G):
    import pytest
    # tests/test_kdtree.py:167-175
    for lat, lon in zip([91, -91, np.nan, 0], [0, 0, 0, np.nan]):
        with pytest.raises(ValueError):
            g.KDTree([lat], [lon], g.Geodetic)
    # tests/test_kdtree.py:177-186
    for lat in (90.000001, -90.0000001):
        tree = g.KDTree([0, lat], [0, 0], g.Geodetic)
        assert tree.get_nearest_neighbour(0, 0) == 0


def pin_nearest(g, G):
    e = G["nearest_grid_to_point"]
    lons, lats = np.meshgrid(e["grid_lons"], e["grid_lats"])
    grid = g.Grid(lats, lons)
    values = np.reshape(np.arange(9), lons.shape)
    points = g.Points(e["point_lats"], e["point_lons"])
    np.testing.assert_array_equal(g.nearest(grid, points, values), e["expected"])
    e1 = G["nearest_one_row"]
    lons, lats = np.meshgrid(e1["grid_lons"], e1["grid_lats"])
    grid = g.Grid(lats, lons)
    values = np.reshape(np.arange(3), lons.shape).astype(float)
    np.testing.assert_array_equal(g.nearest(grid, points, values), e1["expected"])


def _rc(key):
    i, j = key.split(",")
    return int(i), int(j)


def pin_neighbourhood(g, G, funcs=None):
    values = _values5(G)
    funcs = funcs or [g.neighbourhood, g.neighbourhood_brute_force]
    for func in funcs:
        e = G["neighbourhood_mean"]
        out = np.asarray(func(values, 1, g.Mean))
        assert out.dtype == np.float32
        assert out[2][2] == e["hw1"]["2,2"]
        assert abs(out[0][4] - e["hw1"]["0,4"]) < 10 ** -e["hw1_places"]
        out = np.asarray(func(values, 100, g.Mean))
        assert (np.abs(out - e["hw100_all"]) < e["hw100_tol"]).all()
        out = np.asarray(func(values, 0, g.Mean)).flatten()
        assert (np.isnan(out) == np.isnan(values.flatten())).all()
        I = np.where(~np.isnan(out))[0]
        assert (out[I] == values.flatten()[I]).all()

        e = G["neighbourhood_count"]
        out = np.asarray(func(values, 1, g.Count))
        for k, v in e["hw1"].items():
            assert out[_rc(k)] == v
        assert (np.abs(np.asarray(func(values, 100, g.Count)) - e["hw100_all"]) < 1e-4).all()
        np.testing.assert_array_almost_equal(func(values, 0, g.Count), e["hw0"])

        for stat, key in ((g.Min, "neighbourhood_min"), (g.Max, "neighbourhood_max")):
            e = G[key]
            out = np.asarray(func(values, 1, stat))
            for k, v in e["hw1"].items():
                assert out[_rc(k)] == v
            assert (np.asarray(func(values, 100, stat)) == e["hw100_all"]).all()

        # tests/test_neighbourhood.py:48-59
        empty = np.zeros([5, 5])
        empty[0:3, 0:3] = np.nan
        for stat in (g.Mean, g.Min, g.Max, g.Median, g.Std, g.Variance):
            out = np.asarray(func(empty, 1, stat))
            assert np.isnan(out[0:2, 0:2]).all()
        np.testing.assert_array_almost_equal(func(empty, 1, g.Count), G["neighbourhood_missing_count"]["expected"])


def pin_neighbourhood_invalid(g, G):
    import pytest
    field = np.ones([5, 5])
    for stat in (g.Mean, g.Min, g.Max, g.Median):
        with pytest.raises(ValueError):
            g.neighbourhood(field, -1, stat)
    with pytest.raises(Exception):
        g.neighbourhood(field, 1, g.Quantile)
    for stat in (g.Mean, g.Min, g.Max, g.Median, g.Std, g.Variance):
        for func in (g.neighbourhood, g.neighbourhood_brute_force):
            out = np.asarray(func([[]], 1, stat))
            assert out.ndim == 2 and out.shape[0] == 0 and out.shape[1] == 0


def pin_neighbourhood_3d_and_overflow(g, G):
    # tests/test_neighbourhood.py:134-152
    rng = np.random.RandomState(1000)
    values = rng.rand(200, 200)
    values3 = np.repeat(values[:, :, None], 5, axis=2)
    for hw in (0, 1, 5):
        for func in (g.neighbourhood,):
            o2 = func(values, hw, g.Mean)
            o3 = func(values3, hw, g.Mean)
            np.testing.assert_array_almost_equal(o2, o3, 5)
    N = 1000
    values = np.expand_dims(np.arange(1, N) ** 3, 1).astype(float)
    out = np.asarray(g.neighbourhood(values, 0, g.Mean))
    np.testing.assert_array_almost_equal(np.zeros(values.shape), out / values - 1, 6)


def pin_neighbourhood_quantile(g, G):
    import pytest
    values = _values5(G)
    e = G["neighbourhood_quantile"]
    out = np.asarray(g.neighbourhood_quantile(values, e["q"], e["hw"]))
    for k, v in e["expected"].items():
        assert out[_rc(k)] == v
    for q in (-0.1, 1.1):
        with pytest.raises(ValueError):
            g.neighbourhood_quantile(np.ones([5, 5]), q, 1)
    empty = np.zeros([5, 5])
    empty[0:3, 0:3] = np.nan
    assert np.isnan(np.asarray(g.neighbourhood_quantile(empty, 0.5, 1))[0:2, 0:2]).all()


def pin_neighbourhood_quantile_fast(g, G):
    import pytest
    values = _values5(G)
    e = G["neighbourhood_quantile_fast"]
    thresholds = g.get_neighbourhood_thresholds(values, e["num_thresholds"])
    out = np.asarray(g.neighbourhood_quantile_fast(values, e["q"], e["hw"], thresholds))
    for k, v in e["expected"].items():
        assert out[_rc(k)] == v, (k, out[_rc(k)], v)
    out = np.asarray(g.neighbourhood_quantile_fast(np.full([100, 100], np.nan), 0.5, 1, thresholds))
    assert np.isnan(out).all()
    out = np.asarray(g.neighbourhood_quantile_fast(np.zeros([100, 100]), 0.5, 1, thresholds))
    assert (out == 0).all()
    # nan quantile -> nan field (tests/test_neighbourhood_quantile_fast.py:34-41)
    out = np.asarray(g.neighbourhood_quantile_fast(np.ones([5, 5]), np.nan, 1, [0, 1]))
    assert np.isnan(out).all() and out.shape == (5, 5)
    for q in (-0.1, 1.1):
        with pytest.raises(ValueError):
            g.neighbourhood_quantile_fast(np.ones([5, 5]), q, 1, [0, 1])
    # single threshold (:50-56)
    field = np.reshape(np.arange(9), [3, 3])
    for hw in (0, 1, 2):
        np.testing.assert_array_equal(g.neighbourhood_quantile_fast(field, 0.9, hw, [0]), np.zeros([3, 3]))
    # all same (:128-136)
    a = G["quantile_fast_all_same"]
    field = np.zeros([10, 10])
    for q in a["quantiles"]:
        np.testing.assert_array_almost_equal(g.neighbourhood_quantile_fast(field, q, a["hw"], a["thresholds"]), field)
    # 2-D == 3-D (:89-101)
    rng = np.random.RandomState(1000)
    values = rng.rand(200, 200)
    values3 = np.repeat(values[:, :, None], 5, axis=2)
    for hw in (0, 1, 5):
        o2 = g.neighbourhood_quantile_fast(values, 0.5, hw, [0, 0.25, 0.5, 0.75, 1])
        o3 = g.neighbourhood_quantile_fast(values3, 0.5, hw, [0, 0.25, 0.5, 0.75, 1])
        np.testing.assert_array_almost_equal(o2, o3)
    # varying quantile (:103-125)
    v = np.array([[0, 1], [2, 3], [4, 5]], float)
    q = np.ones(v.shape) * 0.5
    g.neighbourhood_quantile_fast(v, q, 1, [0, 0.25, 0.5, 0.75, 1])
    vn = np.nan * np.zeros(v.shape)
    np.testing.assert_array_equal(vn, g.neighbourhood_quantile_fast(vn, q, 1, [0, 0.25, 0.5, 0.75, 1]))


def pin_thresholds(g, G):
    import pytest
    # tests/test_get_neighbourhood_thresholds.py:8-49
    for num in (-1, 0):
        with pytest.raises(ValueError):
            g.get_neighbourhood_thresholds(np.ones([5, 5]), num)
    field = np.reshape(np.arange(4), [2, 2])
    for num in (4, 5, 6):
        np.testing.assert_array_equal(g.get_neighbourhood_thresholds(field, num), [0, 1, 2, 3])
    rng = np.random.RandomState(1000)
    values = rng.rand(10, 10)
    values3 = np.repeat(values[:, :, None], 5, axis=2)
    for num in (1, 5):
        np.testing.assert_array_almost_equal(g.get_neighbourhood_thresholds(values, num),
                                             g.get_neighbourhood_thresholds(values3, num))


def pin_util(g, G):
    import pytest
    stat = dict(Mean=g.Mean, Count=g.Count, Sum=g.Sum, Min=g.Min)
    for c in G["calc_statistic"]["cases"]:
        r = g.calc_statistic(np.array(c["a"], np.float32), stat[c["stat"]])
        if isinstance(c["expected"], float) and np.isnan(c["expected"]):
            assert np.isnan(r)
        else:
            assert r == c["expected"], c
    for c in G["calc_quantile"]["cases"]:
        r = g.calc_quantile(np.array(c["a"], np.float32), c["q"])
        if isinstance(c["expected"], float) and np.isnan(c["expected"]):
            assert np.isnan(r), c
        else:
            assert r == c["expected"], c
    for q in G["calc_quantile"]["invalid_quantiles"]:
        with pytest.raises(ValueError):
            g.calc_quantile(np.array([0, 1, 2], np.float32), q)
    for c in G["calc_even_quantiles"]["cases"]:
        np.testing.assert_array_almost_equal(g.calc_even_quantiles(np.array(c["values"], np.float32), c["num"]), c["expected"])
    for v in G["is_valid"]["valid"]:
        assert g.is_valid(v)
    assert not g.is_valid(np.nan)


def pin_structures(g, G):
    """tests/test_barnes_structure.py:8-33 (Cressman, CrossValidation rows) and tests/test_structure.py:63-154"""
    import pytest
    e = G["structures"]
    x = e["x"]
    barnes = g.BarnesStructure(e["h"])
    table = [(g.CressmanStructure(e["h"]), e["cressman"], False), (g.CrossValidation(barnes, e["cv_dist"]), e["cv"], True)]
    for structure, corr, is_cv in table:
        for xi, c in zip(x, corr):
            p1 = g.Point(0, 0, 0, 0, g.Cartesian)
            p2 = g.Point(xi, 0, 0, 0, g.Cartesian)
            funcs = [structure.corr_background] if is_cv else [structure.corr, structure.corr_background]
            for f in funcs:
                assert abs(f(p1, p2) - c) < 1e-7, (type(structure).__name__, xi, f(p1, p2), c)
                assert abs(f(p2, p1) - c) < 1e-7
    for dist in (-1, np.nan):
        with pytest.raises(Exception):
            g.CrossValidation(barnes, dist)
    # MultipleStructure (tests/test_structure.py:63-91)
    s = g.MultipleStructure(g.CressmanStructure(2000, 2000, 2000), g.CressmanStructure(200, 200, 200), g.CressmanStructure(2, 2, 2))
    p0 = g.Point(0, 0, 0, 0, g.Cartesian)
    for args, expo in (((1000, 0, 0, 0), 1), ((0, 0, 100, 0), 1), ((0, 0, 0, 1), 1), ((1000, 0, 100, 1), 3)):
        assert abs(s.corr(p0, g.Point(*args, g.Cartesian)) - 0.6 ** expo) < 1e-6
    # corr(vec) through OI (tests/test_structure.py:93-124)
    s = g.MultipleStructure(g.CressmanStructure(5000, 11, 22), g.CressmanStructure(33, 200, 44), g.CressmanStructure(55, 66, 2))
    assert abs(s.corr(p0, g.Point(0, 2500, 0, 0, g.Cartesian)) - 0.6) < 1e-6
    assert abs(s.corr(p0, g.Point(0, 2500, 100, 1, g.Cartesian)) - 0.6 ** 3) < 1e-6
    grid = g.Points([0, 0, 0], [0, 0, 0], [0, 0, 100], [0, 0, 1], g.Cartesian)
    points = g.Points([0], [2500], [0], [0], g.Cartesian)
    out = g.optimal_interpolation(grid, np.zeros(3), points, [1], [1], [0], s, 10)
    np.testing.assert_array_almost_equal(out, [0.3, 0.3, 0.6 ** 3 / 2])
    # clone keeps the correlations (tests/test_structure.py:135-154)
    h, v, w = 850, 92, 0.44
    structures = [g.BarnesStructure(h, v, w), g.CressmanStructure(h, v, w),
                  g.MultipleStructure(g.BarnesStructure(1.3 * h, v, w), g.BarnesStructure(h, 1.3 * v, w), g.BarnesStructure(h, v, 1.3 * w)),
                  g.CrossValidation(g.BarnesStructure(h, v, w), 1000)]
    p1, p2 = g.Point(0, 0, 0, 0, g.Cartesian), g.Point(500, 0, 50, 0.25, g.Cartesian)
    for st in structures:
        c = st.clone()
        assert st.corr(p1, p2) == c.corr(p1, p2)
        assert st.corr_background(p1, p2) == c.corr_background(p1, p2)


def pin_oi_cross_validation(g, G):
    """tests/test_optimal_interpolation.py:127-152: the CrossValidation structure equals leave-one-out"""
    y, x = np.meshgrid(np.arange(0, 3500, 500), np.arange(0, 3500, 500))
    grid = g.Grid(y, x, np.zeros(x.shape), np.zeros(x.shape), g.Cartesian)
    background = np.zeros(y.shape)
    obs = np.array([10, 20, 30])
    x_o = np.array([1000, 2000, 3000])
    y_o = np.array([1000, 2000, 3000])
    N = len(obs)
    points = g.Points(y_o, x_o, np.zeros(N), np.zeros(N), g.Cartesian)
    background_o = np.asarray(g.nearest(grid, points, background))
    ratios = np.ones(N)
    k = 0
    ii = np.arange(N) != k
    points_cv = g.Points(y_o[ii], x_o[ii], np.zeros(N - 1), np.zeros(N - 1), g.Cartesian)
    structure = g.BarnesStructure(1000, 0)
    structure_cv = g.CrossValidation(structure, 750)
    analysis = g.optimal_interpolation(grid, background, points_cv, obs[ii], ratios[ii], background_o[ii], structure, 100)
    analysis_cv = g.optimal_interpolation(points, background_o, points, obs, ratios, background_o, structure_cv, 100)
    assert abs(float(np.asarray(g.nearest(grid, points, analysis))[k]) - float(np.asarray(analysis_cv)[k])) < 1e-6


def pin_nearest_overloads(g, G):
    e = G["nearest_overloads"]
    c = e["grid_to_grid"]
    lons1, lats1 = np.meshgrid(c["in_lons"], c["in_lats"])
    lons2, lats2 = np.meshgrid(c["out_lons"], c["out_lats"])
    grid1, grid2 = g.Grid(lats1, lons1), g.Grid(lats2, lons2)
    np.testing.assert_array_equal(g.nearest(grid1, grid2, np.reshape(range(9), lons1.shape)), c["expected"])
    np.testing.assert_array_equal(g.nearest(grid1, grid2, np.reshape(range(18), (2,) + lons1.shape)), c["expected_3d"])
    c = e["grid_to_points_3d"]
    lons, lats = np.meshgrid(c["axis"], c["axis"])
    out = g.nearest(g.Grid(lats, lons), g.Points(c["point_lats"], c["point_lons"]), np.reshape(range(18), (2,) + lons.shape))
    np.testing.assert_array_equal(out, c["expected"])
    c = e["points_to_points"]
    ip, op = g.Points(c["in"], c["in"]), g.Points(c["out"], c["out"])
    np.testing.assert_array_equal(g.nearest(ip, op, c["values"]), c["expected"])
    np.testing.assert_array_equal(g.nearest(ip, op, c["values_2d"]), c["expected_2d"])
    np.testing.assert_array_equal(g.nearest(ip, op, [c["values"]]), [c["expected"]])
    # points -> grid (src/api/nearest.cpp:73-122)
    out = g.nearest(ip, grid2, c["values"])
    assert np.shape(out) == (2, 2)
    out = g.nearest(ip, grid2, c["values_2d"])
    assert np.shape(out) == (2, 2, 2)
    # empty inputs / outputs (tests/test_nearest.py:114-146)
    out = g.nearest(g.Points([], []), g.Points([0, 5, 10], [0, 5, 10]), np.zeros([0]))
    assert np.size(out) == 3 and np.all(np.isnan(np.asarray(out)))
    assert np.size(g.nearest(ip, g.Points([], []), np.zeros([3]))) == 0
    out = g.nearest(g.Grid([[]], [[]]), g.Points([0, 5, 10], [0, 5, 10]), np.zeros([0, 0]))
    assert np.size(out) == 3 and np.all(np.isnan(np.asarray(out)))


def _nan(rows):
    return np.array([[np.nan if v is None else v for v in r] for r in rows], np.float32)


def pin_gridding(g, G):
    import pytest
    e = G["gridding"]
    y, x = np.meshgrid(e["grid_y_axis"], e["grid_x_axis"])
    grid = g.Grid(y, x, 0 * y, 0 * y, g.Cartesian)
    grid_as_points = grid.to_points()
    pts = g.Points(e["points"], e["points"], [0, 0, 0], [0, 0, 0], g.Cartesian)
    values, radius = e["values"], e["radius"]
    for target in (grid, grid_as_points):
        shape = tuple(target.size()) if isinstance(target.size(), (list, tuple)) else (target.size(),)
        for min_num, expected in e["min_num"].items():
            out = np.asarray(g.gridding(target, pts, values, radius, int(min_num), g.Sum))
            assert out.shape == shape
            np.testing.assert_array_almost_equal(out.ravel(), _nan(expected).ravel())
        for name, expected in e["statistic"].items():
            out = np.asarray(g.gridding(target, pts, values, radius, 0, getattr(g, name)))
            np.testing.assert_array_almost_equal(out.ravel(), _nan(expected).ravel())
        for r, expected in e["radius_cases"].items():
            out = np.asarray(g.gridding(target, pts, values, float(r), 0, g.Sum))
            np.testing.assert_array_almost_equal(out.ravel(), _nan(expected).ravel())
        with pytest.raises(ValueError):
            g.gridding(target, pts, [0], radius, 0, g.Sum)
        for bad in (-1, np.nan):
            with pytest.raises(ValueError):
                g.gridding(target, pts, values, bad, 0, g.Sum)
        with pytest.raises(ValueError):
            g.gridding(target, pts, values, radius, -1, g.Sum)
        # empty input points: NaN, Count 0 (tests/test_gridding.py:79-94)
        empty = g.Points([], [], [], [], g.Cartesian)
        for stat in (g.Sum, g.Mean):
            out = np.asarray(g.gridding(target, empty, [], radius, 0, stat))
            assert out.shape == shape and np.all(np.isnan(out))
        np.testing.assert_array_equal(np.asarray(g.gridding(target, empty, [], radius, 0, g.Count)).ravel(), np.zeros(6))
    # empty outputs (tests/test_gridding.py:96-106)
    for stat in (g.Sum, g.Mean, g.Count):
        assert np.size(g.gridding(g.Grid([[]], [[]], [[]], [[]], g.Cartesian), pts, values, radius, 0, stat)) == 0
        assert np.size(g.gridding(g.Points([], [], [], [], g.Cartesian), pts, values, radius, 0, stat)) == 0


def pin_count(g, G):
    e = G["count"]
    for name, factor in e["factor"].items():
        ct = getattr(g, name)
        lons, lats = np.meshgrid(e["grid_lons"], e["grid_lats"])
        grid = g.Grid(lats * factor, lons * factor, lons * 0, lons * 0, ct)
        plats, plons = np.array(e["point_lats"]) * factor, np.array(e["point_lons"]) * factor
        points = g.Points(plats, plons, plons * 0, plons * 0, ct)
        radius = e["radius"]
        single_grid = g.Grid([[0]], [[0]], [[0]], [[0]], ct)
        single_point = g.Points([0], [0], [0], [0], ct)
        empty_grid = g.Grid([[]], [[]], [[]], [[]], ct)
        empty_points = g.Points([], [], [], [], ct)
        np.testing.assert_array_equal(g.count(points, grid, radius), e["point_to_grid"])
        assert np.size(g.count(points, empty_grid, radius)) == 0
        np.testing.assert_array_equal(g.count(empty_points, grid, radius), np.zeros((3, 2)))
        np.testing.assert_array_equal(g.count(grid, grid, radius), e["grid_to_grid"])
        np.testing.assert_array_equal(g.count(grid, single_grid, radius), e["grid_to_single_grid"])
        np.testing.assert_array_equal(g.count(single_grid, grid, radius), e["single_grid_to_grid"])
        np.testing.assert_array_equal(g.count(empty_grid, grid, radius), np.zeros((3, 2)))
        assert np.size(g.count(grid, empty_grid, radius)) == 0
        np.testing.assert_array_equal(g.count(grid, points, radius), e["grid_to_point"])
        np.testing.assert_array_equal(g.count(single_grid, points, radius), e["single_grid_to_point"])
        np.testing.assert_array_equal(g.count(grid, single_point, radius), e["grid_to_single_point"])
        np.testing.assert_array_equal(g.count(empty_grid, points, radius), [0, 0, 0])
        assert np.size(g.count(grid, empty_points, radius)) == 0
        np.testing.assert_array_equal(g.count(points, points, radius), e["point_to_point"])
        assert np.size(g.count(points, empty_points, radius)) == 0
        np.testing.assert_array_equal(g.count(empty_points, points, radius), [0, 0, 0])


def pin_fill(g, G):
    import pytest
    e = G["fill"]
    lons, lats = np.meshgrid(e["axis"], e["axis"])
    grid = g.Grid(lats * e["scale"], lons * e["scale"], np.zeros([5, 5]), np.zeros([5, 5]), g.Cartesian)
    points = g.Points(e["point_lats"], e["point_lons"], [0, 0, 0], [0, 0, 0], g.Cartesian)
    values = np.zeros([5, 5], np.float32)
    np.testing.assert_array_equal(g.fill(grid, values, points, e["radii"], e["value"], False), e["inside"])
    np.testing.assert_array_equal(g.fill(grid, values, points, e["radii"], e["value"], True), e["outside"])
    for outside in (False, True):       # tests/test_fill.py:10-31
        with pytest.raises(Exception):
            g.fill(grid, values, points, [-1, -1, -1], 1, outside)
        with pytest.raises(Exception):
            g.fill(grid, values, points, [1], 1, outside)
        with pytest.raises(Exception):
            g.fill(grid, np.zeros([3, 2], np.float32), points, e["radii"], 1, outside)
    e = G["fill_missing"]
    for c in e["cases"]:
        n = c["shape"][0] * c["shape"][1]
        values0 = np.reshape(np.arange(n), c["shape"]).astype(np.float32)
        values = values0.copy()
        for i, j in c["nan"]:
            values[i, j] = np.nan
        out = np.asarray(g.fill_missing(values))
        if c["full"]:
            np.testing.assert_array_equal(out, values0)
        else:                            # tests/test_fill_missing.py:41-48
            np.testing.assert_array_equal(out[0:3, :], values0[0:3, :])
            np.testing.assert_array_equal(out[:, 0:3], values0[:, 0:3])
            assert np.all(np.isnan(out[3:5, 3:5]))


def pin_doping(g, G):
    e = G["doping_square"]
    N = e["N"]
    x = np.linspace(0, e["extent"], N)
    xx, yy = np.meshgrid(x, x)
    grid = g.Grid(xx, yy, 0 * xx, 0 * xx, g.Cartesian)
    points = g.Points(e["point_lats"], e["point_lons"], [0, 0], [0, 0], g.Cartesian)
    out = g.doping_square(grid, np.zeros([N, N], np.float32), points, e["obs"], e["halfwidth"], e["max_elev_diff"])
    expected = np.zeros([N, N])
    for sq in e["squares"]:
        expected[sq["rows"][0]:sq["rows"][1], sq["cols"][0]:sq["cols"][1]] = sq["value"]
    np.testing.assert_array_almost_equal(out, expected)


def pin_neighbourhood_search(g, G):
    e = G["neighbourhood_search"]
    nan = lambda a: np.array([[np.nan if v is None else v for v in r] for r in a], np.float32)
    apply = lambda base, lo, hi: ((nan(base) >= lo) & (nan(base) <= hi)).astype(np.int32)
    a = e["args"]
    np.testing.assert_array_equal(g.neighbourhood_search(e["values"], e["base"], a[0], a[1], a[2], a[3], apply(e["base"], 0, 0.95)), e["results"])
    np.testing.assert_array_equal(g.neighbourhood_search(e["values"], e["base"], a[0], a[1], a[2], a[3]), e["results_no_apply"])
    np.testing.assert_array_equal(g.neighbourhood_search(e["values2"], e["base2"], a[0], a[1], a[2], a[3], apply(e["base2"], 0, 0.85)), e["values2"])
    np.testing.assert_array_equal(g.neighbourhood_search(nan(e["values_nan"]), nan(e["base_nan"]), a[0], a[1], a[2], a[3], apply(e["base_nan"], 0, 0.95)),
                                  e["results_nan"])
    s_ = e["simple"]
    np.testing.assert_array_equal(g.neighbourhood_search(s_["array"], s_["search"], *s_["args"]), s_["expected"])


def pin_calc_gradient(g, G):
    import pytest
    e = G["calc_gradient"]
    row = lambda a: np.array([[np.nan if v is None else v for v in a]], np.float32)
    c = e["simple"]
    out = g.calc_gradient(row(c["base"]), row(c["values"]), g.LinearRegression, c["halfwidth"], c["min_num"], c["min_range"], c["default"])
    np.testing.assert_array_almost_equal(out, [c["expected"]])
    c2 = e["small"]
    np.testing.assert_array_almost_equal(g.calc_gradient(row(c2["base"]), row(c2["values"]), g.LinearRegression, c2["halfwidth"], 0, 0, -11), [c2["expected"]])
    c3 = e["num_min"]
    np.testing.assert_array_almost_equal(g.calc_gradient(row(c3["base"]), row(c3["values"]), g.LinearRegression, c3["halfwidth"], c3["min_num"], 0, -11),
                                         [c3["expected"]])
    z = np.zeros([3, 2], np.float32)      # tests/test_calc_gradient.py:40-56
    with pytest.raises(ValueError):
        g.calc_gradient(z, np.zeros([2, 3], np.float32), g.LinearRegression, 5, 0, 0, -11)
    for hw, mn, mr in ((-1, 0, 0), (5, -1, 0), (5, 0, -1)):
        with pytest.raises(ValueError):
            g.calc_gradient(z, z, g.LinearRegression, hw, mn, mr, -11)


def pin_distance(g, G):
    e = G["distance"]["cartesian"]
    lons, lats = np.meshgrid(e["grid_lons"], e["grid_lats"])
    grid = g.Grid(lats, lons, 0 * lats, 0 * lats, g.Cartesian)
    points = g.Points(e["point_lats"], e["point_lons"], [0, 0], [0, 0], g.Cartesian)
    for num, expected in e["point_to_grid"].items():
        np.testing.assert_array_almost_equal(g.distance(points, grid, int(num)), expected, e["places"])
    for num, expected in e["grid_to_point"].items():
        np.testing.assert_array_almost_equal(g.distance(grid, points, int(num)), expected, e["places"])
    e = G["distance"]["geodetic"]
    lons, lats = np.meshgrid(e["grid_lons"], e["grid_lats"])
    grid = g.Grid(lats, lons)
    points = g.Points(e["point_lats"], e["point_lons"])
    for num, expected in e["grid_to_point"].items():
        np.testing.assert_array_almost_equal(g.distance(grid, points, int(num)), expected, e["places"])


def pin_containers_next(g, G):
    e = G["containers_next"]
    c = e["in_domain"]
    lons, lats = np.meshgrid(c["axis"], c["axis"])
    grid = g.Grid(lats, lons)
    points = g.Points(c["lats"], c["lons"])
    np.testing.assert_array_equal(points.get_in_domain_indices(grid), c["inside"])
    sub = points.get_in_domain(grid)
    np.testing.assert_array_equal(np.sort(sub.get_lats()), np.sort(np.array(c["lats"], np.float32)[c["inside"]]))
    np.testing.assert_array_equal(np.sort(sub.get_lons()), np.sort(np.array(c["lons"], np.float32)[c["inside"]]))
    assert len(g.Points([], []).get_in_domain_indices(grid)) == 0
    assert len(g.Points([0], [0]).get_in_domain_indices(g.Grid())) == 0
    c = e["grid_with_distance"]
    cg = g.Grid(c["lats"], c["lons"], np.zeros([3, 2]), np.zeros([3, 2]), g.Cartesian)
    indices, distances = cg.get_neighbours_with_distance(*c["q"])
    assert len(indices) == 4
    np.testing.assert_array_almost_equal(np.sort(distances), c["distances"], 4)
    c = e["grid_get_point"]
    p = g.Grid(c["lats"], c["lons"], c["elevs"], c["lafs"]).get_point(*c["yx"])
    for k, v in c["expected"].items():
        assert getattr(p, k) == v
    ok, x, y, z = g.convert_coordinates(c["expected"]["lat"], c["expected"]["lon"], g.Geodetic)
    assert (p.x, p.y, p.z) == (x, y, z)
    c = e["distance_limit"]
    p0, p1 = g.Point(c["args"][0], c["args"][1]), g.Point(c["args"][2], c["args"][3])
    assert abs(g.KDTree_calc_distance(*c["args"]) - c["expected"]) < 5e-8      # assertAlmostEqual: 7 places
    assert abs(g.KDTree_calc_straight_distance(p0.x, p0.y, p0.z, p1.x, p1.y, p1.z) - c["expected"]) < 5e-8
    for rad, deg in e["rad2deg"]:
        assert abs(g.KDTree_rad2deg(rad) - deg) < 1e-4


def pin_bilinear(g, G):
    import pytest
    e = G["bilinear_simple"]
    lons, lats = np.meshgrid(e["grid_lons"], e["grid_lats"])
    grid = g.Grid(lats, lons)
    values = np.reshape(np.arange(9), lons.shape).astype(np.float32)
    points = g.Points(e["point_lats"], e["point_lons"])
    np.testing.assert_array_equal(g.bilinear(grid, points, values), e["expected"])
    # incompatible sizes (tests/test_bilinear.py:29-40,158-163)
    grid24 = g.Grid([[0, 0, 0, 0], [1, 1, 1, 1]], [[0, 1, 2, 3], [0, 1, 2, 3]])
    for N in (1, 3, 5):
        with pytest.raises(Exception):
            g.bilinear(grid24, grid24, np.zeros([N, N], np.float32))
        with pytest.raises(Exception):
            g.bilinear(grid24, g.Points([0, 1], [0, 1]), np.zeros([N, N], np.float32))
    lons2, lats2 = np.meshgrid([0, 1], [0, 1])
    with pytest.raises(Exception):
        g.bilinear(g.Grid(lats2, lons2), g.Grid(lats2, lons2), np.zeros([3, 1, 1], np.float32))
    # empty outputs / inputs (tests/test_bilinear.py:165-187)
    assert np.size(g.bilinear(grid, g.Points([], []), values)) == 0
    out = g.bilinear(g.Grid([[]], [[]]), g.Points([0, 5, 10], [0, 5, 10]), np.zeros([0, 0], np.float32))
    assert np.all(np.isnan(np.asarray(out))) and np.size(out) == 3
    assert np.size(g.bilinear(grid, g.Grid([[]], [[]]), values)) == 0


def pin_bilinear_shapes(g, G):
    e = G["bilinear_rotation"]
    values = np.array(e["values"], np.float32)
    lons0, lats0 = np.array(e["lons0"], float), np.array(e["lats0"], float)
    for rotation in e["rotations_deg"]:
        angle = rotation * 2 * e["pi"] / 360
        lon_p = e["point_lon0"] * np.cos(angle) - e["point_lat0"] * np.sin(angle)
        lat_p = e["point_lon0"] * np.sin(angle) + e["point_lat0"] * np.cos(angle)
        lons = lons0 * np.cos(angle) - lats0 * np.sin(angle)
        lats = lons0 * np.sin(angle) + lats0 * np.cos(angle)
        out = g.bilinear(g.Grid(lats, lons), g.Points([lat_p], [lon_p]), values)[0]
        assert abs(float(out) - e["expected"]) < 0.5 * 10 ** -e["places"], (rotation, out)
    e = G["bilinear_parallelogram"]
    values = np.array(e["values"], np.float32)
    origin = g.Points([0], [0])
    for skew in e["skews"]:
        grid = g.Grid([[-1, -1], [1, 1]], [[-1 + skew, 1 + skew], [-1 - skew, 1 - skew]])
        assert g.bilinear(grid, origin, values)[0] == e["expected"]
        grid = g.Grid([[-1 + skew, -1 - skew], [1 + skew, 1 - skew]], [[-1, 1], [-1, 1]])
        assert g.bilinear(grid, origin, values)[0] == e["expected"]
    e = G["bilinear_non_parallelogram"]
    out = g.bilinear(g.Grid(e["lats"], e["lons"]), g.Points([e["point"][0]], [e["point"][1]]), np.array(e["values"], np.float32))[0]
    assert abs(float(out) - e["expected"]) < 0.5 * 10 ** -e["places"]
    e = G["bilinear_vertical_parallel"]          # must not fail
    for c in e["cases"]:
        lons = c["lons"] if "lons" in c else np.transpose(c["lons_T"])
        lats = c["lats"] if "lats" in c else np.transpose(c["lats_T"])
        out = g.bilinear(g.Grid(lats, lons), g.Points([e["point"][0]], [e["point"][1]]), np.array(e["values"], np.float32))
        assert np.size(out) == 1
    e = G["bilinear_weird"]
    x = np.reshape(e["x"], [2, 2]).transpose()
    y = np.reshape(e["y"], [2, 2]).transpose()
    x0, y0 = e["px"] - x[0][0], e["py"] - y[0][0]
    x, y = x - x[0][0], y - y[0][0]
    values = np.reshape(np.arange(4), [2, 2]).transpose().astype(np.float32)
    for _ in range(2):
        x, y, values = x.transpose(), y.transpose(), values.transpose()
        q = g.bilinear(g.Grid(y, x), g.Points([y0], [x0]), values)
        assert abs(float(q[0]) - e["expected"]) < 0.5 * 10 ** -e["places"], q


def pin_bilinear_missing_and_grids(g, G):
    e = G["bilinear_missing"]
    values = np.array([[np.nan if v is None else v for v in row] for row in e["values"]], np.float32)
    out = g.bilinear(g.Grid(e["lats"], e["lons"]), g.Points(e["point_lats"], e["point_lons"]), values)
    np.testing.assert_array_equal(out, e["expected"])
    e = G["bilinear_grid_to_grid"]
    values = np.reshape(np.arange(4), [2, 2]).astype(np.float32)
    lons1, lats1 = np.meshgrid(e["in_axis"], e["in_axis"])
    lons2, lats2 = np.meshgrid(e["out_axis"], e["out_axis"])
    grid1, grid2 = g.Grid(lats1, lons1), g.Grid(lats2, lons2)
    np.testing.assert_array_equal(g.bilinear(grid1, grid2, values), e["expected"])
    out = np.asarray(g.bilinear(grid1, grid2, np.repeat(values[None], e["T"], axis=0)))
    assert out.shape == (e["T"], 3, 3)
    for t in range(e["T"]):
        np.testing.assert_array_equal(out[t], e["expected"])


def pin_grid_get_box(g, G):
    e = G["grid_get_box"]
    grid = g.Grid(e["lats"], e["lons"])
    for c in e["cases"]:
        np.testing.assert_array_equal(grid.get_box(*c["q"]), c["expected"])
    o = e["one_row"]
    assert not g.Grid(o["lats"], o["lons"]).get_box(*o["q"])[0]
    assert not g.Grid().get_box(*o["q"])[0]


def pin_point_in_rectangle(g, G):
    for c in G["point_in_rectangle"]["cases"]:
        A, B, C, D = (g.Point(*c[k]) for k in "ABCD")
        for m in c["inside"]:
            assert g.point_in_rectangle(A, B, C, D, g.Point(*m)), (c, m)
        for m in c["outside"]:
            assert not g.point_in_rectangle(A, B, C, D, g.Point(*m)), (c, m)


ALL_PINS = [v for k, v in sorted(globals().items()) if k.startswith("pin_")]
