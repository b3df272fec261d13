"""A gridpp-shaped facade over the CPU oracle, so that the reference-style pins
in tests/pins.py run unchanged against the oracle (CPU) and gridpp_amd (GPU)."""
import numpy as np

from oracle import oracle as O

Geodetic, Cartesian = 0, 1
Mean, Min, Median, Max, Quantile, Std, Variance, Sum, Count, RandomChoice = 0, 10, 20, 30, 40, 50, 60, 70, 80, 90
name = "oracle"


class Point:
    def __init__(self, lat, lon, elev=np.nan, laf=np.nan, type=Geodetic):
        self.lat, self.lon, self.elev, self.laf, self.type = lat, lon, elev, laf, type
        if type == Geodetic:
            x, y, z = O.convert_coordinates([lat], [lon], type)
            self.x, self.y, self.z = float(x[0]), float(y[0]), float(z[0])
        else:  # src/api/point.cpp:18-21
            self.x, self.y, self.z = float(np.float32(lat)), float(np.float32(lon)), 0.0


class Points:
    def __init__(self, lats=(), lons=(), elevs=(), lafs=(), type=Geodetic):
        lats, lons = np.asarray(lats, np.float32).ravel(), np.asarray(lons, np.float32).ravel()
        if lats.size != lons.size:
            raise ValueError("size")
        try:
            self.p = O.Pts(lats, lons, elevs if np.size(elevs) else None, lafs if np.size(lafs) else None, type)
        except O.OracleError as e:
            raise ValueError(str(e))
        self.type = type

    def size(self):
        return self.p.n

    def get_coordinate_type(self):
        return self.type

    def get_neighbours(self, lat, lon, radius, include_match=True):
        return O.get_neighbours(self.p, lat, lon, radius, include_match)

    def get_nearest_neighbour(self, lat, lon, include_match=True):
        return O.nearest_neighbour(self.p, lat, lon, include_match)

    def get_lats(self):
        return self.p.lats

    def get_lons(self):
        return self.p.lons

    def get_in_domain_indices(self, grid):       # src/api/points.cpp:77-92
        if self.p.n == 0 or grid.p.n == 0:
            return np.zeros(0, np.int32)
        return np.array([s for s in range(self.p.n) if grid.get_box(float(self.p.lats[s]), float(self.p.lons[s]))[0]], np.int32)

    def get_in_domain(self, grid):               # points.cpp:93-109
        i = self.get_in_domain_indices(grid)
        return Points(self.p.lats[i], self.p.lons[i], self.p.elevs[i], self.p.lafs[i])


KDTree = Points


class Grid:
    def __init__(self, lats=((),), lons=((),), elevs=((),), lafs=((),), type=Geodetic):
        lats = np.asarray(lats, np.float32)
        lons = np.asarray(lons, np.float32)
        self.shape = lats.shape if lats.ndim == 2 else (0, 0)
        e = np.asarray(elevs, np.float32)
        l = np.asarray(lafs, np.float32)
        self.p = O.Pts(lats.ravel(), lons.ravel(), e.ravel() if e.size == lats.size and e.size else None,
                       l.ravel() if l.size == lats.size and l.size else None, type)
        self.type = type

    def size(self):
        return list(self.shape)

    def to_points(self):
        return Points(self.p.lats, self.p.lons, self.p.elevs, self.p.lafs, self.type)

    def get_point(self, y, x):                   # src/api/grid.cpp:230-233
        i = y * self.shape[1] + x
        q = Point(self.p.lats[i], self.p.lons[i], self.p.elevs[i], self.p.lafs[i], self.type)
        q.x, q.y, q.z = float(self.p.x[i]), float(self.p.y[i]), float(self.p.z[i])
        return q

    def get_neighbours_with_distance(self, lat, lon, radius, include_match=True):   # grid.cpp:62-66
        idx = O.get_neighbours(self.p, lat, lon, radius, include_match)
        qx, qy, qz = O.convert_coordinates([lat], [lon], self.type)
        d = np.array([O.lib().orc_calc_straight_distance(self.p.x[i], self.p.y[i], self.p.z[i], qx[0], qy[0], qz[0]) for i in idx], np.float32)
        return np.stack([idx // self.shape[1], idx % self.shape[1]], axis=1) if len(idx) else np.zeros((0, 2), np.int32), d

    def get_box(self, lat, lon):
        return O.get_box(self.p, tuple(self.shape) if self.p.n else (0, 0), lat, lon)

    def get_coordinate_type(self):
        return self.type


class _Structure:
    kind = "Barnes"

    def __init__(self, h, v=0, w=0, hmax=np.nan):
        if not np.isfinite(h) or h < 0:
            raise ValueError("h")
        if np.isfinite(hmax) and hmax < 0:
            raise ValueError("hmax")
        self.g = O.Struct(self.kind, h, v, w, hmax)
        # the Barnes-only oracle entry points (orc_oi_full_range, EnSI) take (h, v, w, min_rho)
        self.s = O.Barnes(h, v, w, hmax) if self.kind == "Barnes" else None

    def _p(self, p):
        return (p.x, p.y, p.z, p.elev, p.laf)

    def corr(self, p1, p2):
        return self.g.corr(self._p(p1), self._p(p2), False)

    def corr_background(self, p1, p2):
        return self.g.corr(self._p(p1), self._p(p2), True)

    def localization_distance(self, p=None):
        return self.g.localization_distance()

    def clone(self):
        import copy
        return copy.copy(self)


class BarnesStructure(_Structure):
    kind = "Barnes"


class CressmanStructure(_Structure):
    kind = "Cressman"

    def __init__(self, h, v=0, w=0):
        _Structure.__init__(self, h, v, w)


class SoarStructure(_Structure):
    kind = "Soar"


class ToarStructure(_Structure):
    kind = "Toar"


class PowerlawStructure(_Structure):
    kind = "Powerlaw"


class LinearStructure(_Structure):
    kind = "Linear"


class MultipleStructure(_Structure):
    def __init__(self, sh, sv, sw):
        self.g = O.Struct.multiple(sh.g, sv.g, sw.g)
        self.s = None


class CrossValidation(_Structure):
    def __init__(self, structure, dist):
        if not np.isfinite(dist) or dist < 0:
            raise ValueError("dist")
        self.g = structure.g.cross_validation(dist)
        self.s = None


def _pts(obj):
    return obj.p


def _validate(bg, background, points, max_points, *per_point):
    # src/api/oi.cpp:38-63 / 100-121
    if max_points < 0:
        raise ValueError("max_points must be >= 0")
    if bg.get_coordinate_type() != points.get_coordinate_type():
        raise ValueError("coordinate type mismatch")
    shape = tuple(bg.size()) if isinstance(bg, Grid) else (bg.size(),)
    if tuple(np.shape(background))[:len(shape)] != shape:
        raise ValueError("background size mismatch")
    for a in per_point:
        if np.shape(a)[0] != points.size():
            raise ValueError("points size mismatch")


def optimal_interpolation(bg, background, points, pobs, pratios, pbackground, structure, max_points, allow_extrapolation=True):
    background = np.asarray(background, np.float32)
    _validate(bg, background, points, max_points, pobs, pratios, pbackground)
    if structure.s is None:   # generic structure: the general-inverse oracle
        n = int(np.prod(background.shape))
        out, _ = O.oi_full_generic(_pts(bg), background.ravel(), np.ones(n, np.float32), _pts(points), pobs, pratios, pbackground,
                                   np.ones(points.size(), np.float32), structure.g, max_points, allow_extrapolation)
        return out.reshape(background.shape)
    out = O.oi(_pts(bg), background.ravel(), _pts(points), pobs, pratios, pbackground, structure.s, max_points, allow_extrapolation)
    return out.reshape(background.shape)


def optimal_interpolation_full(bg, background, bvariance, points, pobs, obs_variance, pbackground, bvariance_at_points,
                               structure, max_points, allow_extrapolation=True):
    background = np.asarray(background, np.float32)
    out, var = O.oi_full(_pts(bg), background.ravel(), np.asarray(bvariance, np.float32).ravel(), _pts(points), pobs,
                         obs_variance, pbackground, bvariance_at_points, structure.s, max_points, allow_extrapolation)
    return out.reshape(background.shape), var.reshape(background.shape)


def optimal_interpolation_ensi(bg, background, points, pobs, psigmas, pbackground, structure, max_points, allow_extrapolation=True):
    background = np.asarray(background, np.float32)
    E = background.shape[-1]
    out = O.oi_ensi(_pts(bg), background.reshape(-1, E), _pts(points), pobs, psigmas, pbackground, structure.s, max_points,
                    allow_extrapolation)
    return out.reshape(background.shape)


def nearest(grid, points, values):
    values = np.asarray(values, np.float32)
    nd = 2 if isinstance(grid, Grid) else 1
    oshape = tuple(points.size()) if isinstance(points, Grid) else (points.size(),)
    if values.size == 0 or int(np.prod(oshape)) == 0:
        lead = (values.shape[0],) if values.ndim == nd + 1 else ()
        return np.full(lead + oshape, np.nan, np.float32)
    if values.ndim == nd + 1:
        return np.stack([O.nearest(_pts(grid), _pts(points), v).reshape(oshape) for v in values])
    return O.nearest(_pts(grid), _pts(points), values).reshape(oshape)


def _oshape(o):
    return tuple(o.size()) if isinstance(o, Grid) else (o.size(),)


def count(ipoints, opoints, radius):
    return O.count(_pts(ipoints), _pts(opoints), radius).reshape(_oshape(opoints) if _pts(opoints).n else ((0, 0) if isinstance(opoints, Grid) else (0,)))


def gridding(ogrid, ipoints, values, radius, min_num, statistic):
    if not np.isfinite(radius) or radius < 0:
        raise ValueError("radius must be >= 0")
    if min_num < 0:
        raise ValueError("min_num must be >= 0")
    try:
        out = O.gridding(_pts(ogrid), _pts(ipoints), values, radius, min_num, statistic)
    except O.OracleError as e:
        raise ValueError(str(e))
    return out.reshape(_oshape(ogrid) if _pts(ogrid).n else ((0, 0) if isinstance(ogrid, Grid) else (0,)))


def gridding_nearest(ogrid, ipoints, values, min_num, statistic):
    if min_num < 0:
        raise ValueError("min_num must be >= 0")
    try:
        out = O.gridding_nearest(_pts(ogrid), _pts(ipoints), values, min_num, statistic)
    except O.OracleError as e:
        raise ValueError(str(e))
    return out.reshape(_oshape(ogrid) if _pts(ogrid).n else ((0, 0) if isinstance(ogrid, Grid) else (0,)))


MinMax, LinearRegression = 0, 10


def _grid_values(igrid, values, what="values"):
    values = np.asarray(values, np.float32)
    if values.ndim != 2 or (values.shape[0] and tuple(values.shape) != (tuple(igrid.size()) if igrid.p.n else (0, 0))):
        raise ValueError("Grid size is not the same as " + what)
    return values


def fill(igrid, input, points, radii, value, outside):
    input = _grid_values(igrid, input)
    if np.size(radii) != points.size():
        raise ValueError("Points size is not the same as radii size")
    try:
        return O.fill(_pts(igrid), input, _pts(points), radii, value, outside)
    except O.OracleError as e:
        raise ValueError(str(e))


def fill_missing(values):
    return O.fill_missing(values)


def doping_square(igrid, background, points, observations, halfwidth, max_elev_diff=np.nan):
    background = _grid_values(igrid, background, "observations")
    if np.size(observations) != points.size() or np.size(halfwidth) != points.size():
        raise ValueError("Points size mismatch")
    try:
        return O.doping_square(_pts(igrid), tuple(igrid.size()), background, _pts(points), observations, halfwidth, max_elev_diff)
    except O.OracleError as e:
        raise ValueError(str(e))


def doping_circle(igrid, background, points, observations, radii, max_elev_diff=np.nan):
    background = _grid_values(igrid, background, "observations")
    if np.size(observations) != points.size() or np.size(radii) != points.size():
        raise ValueError("Points size mismatch")
    try:
        return O.doping_circle(_pts(igrid), background, _pts(points), observations, radii, max_elev_diff)
    except O.OracleError as e:
        raise ValueError(str(e))


def neighbourhood_search(array, search_array, halfwidth, search_target_min, search_target_max, search_delta, apply_array=None):
    array, search_array = np.asarray(array, np.float32), np.asarray(search_array, np.float32)
    if array.shape != search_array.shape:
        raise ValueError("search_array must be the same size as array")
    if apply_array is not None and np.size(apply_array) > 1 and np.shape(apply_array) != array.shape:
        raise ValueError("apply_array must either be empty or same size as array")
    if apply_array is not None and np.size(apply_array) == 0:
        apply_array = None
    try:
        return O.neighbourhood_search(array, search_array, halfwidth, search_target_min, search_target_max, search_delta, apply_array)
    except O.OracleError as e:
        raise ValueError(str(e))


def calc_gradient(base, values, gradient_type, halfwidth, num_min=2, min_range=np.nan, default_gradient=0):
    base, values = np.asarray(base, np.float32), np.asarray(values, np.float32)
    if base.size == 0:
        raise ValueError("base input has no size")
    if base.shape != values.shape:
        raise ValueError("base is not the same size as values")
    try:
        return O.calc_gradient(base, values, gradient_type, halfwidth, num_min, min_range, default_gradient)
    except O.OracleError as e:
        raise ValueError(str(e))


def distance(ipoints, opoints, num):
    if ipoints.get_coordinate_type() != opoints.get_coordinate_type():
        raise ValueError("Incompatible coordinate types")
    shape = _oshape(opoints) if _pts(opoints).n else ((0, 0) if isinstance(opoints, Grid) else (0,))
    return O.distance(_pts(ipoints), _pts(opoints), num, not isinstance(opoints, Grid)).reshape(shape)


def bilinear(igrid, opoints, values):
    values = np.asarray(values, np.float32)
    ishape = tuple(igrid.size()) if igrid.p.n else (0, 0)      # src/api/grid.cpp:122-130
    # src/api/util.cpp:427-432: no rows (2-D) / no times or no rows (3-D) is compatible with anything
    empty = values.ndim in (2, 3) and (values.shape[0] == 0 or (values.ndim == 3 and values.shape[1] == 0))
    if values.ndim not in (2, 3) or (not empty and tuple(values.shape[-2:]) != ishape):
        raise ValueError("Grid size is not the same as values")
    oshape = tuple(opoints.size()) if isinstance(opoints, Grid) else (opoints.size(),)
    lead = (values.shape[0],) if values.ndim == 3 else ()
    if int(np.prod(oshape)) == 0:
        return np.zeros(lead + oshape, np.float32)
    if igrid.p.n == 0:                                           # src/api/bilinear.cpp:37-39
        return np.full(lead + oshape, np.nan, np.float32)
    try:
        out = O.bilinear(_pts(igrid), ishape, _pts(opoints), values)
    except O.OracleDistorted as e:
        raise RuntimeError(str(e))
    return out.reshape(lead + oshape)


def point_in_rectangle(A, B, C, D, m):
    return O.point_in_rectangle((A.lat, A.lon), (B.lat, B.lon), (C.lat, C.lon), (D.lat, D.lon), (m.lat, m.lon))


def _wrap(fn):
    def f(*a, **k):
        try:
            return fn(*a, **k)
        except O.OracleError as e:
            raise ValueError(str(e))
    return f


def _empty2(field):
    f = np.asarray(field, np.float32)
    return f.ndim >= 2 and (f.shape[0] == 0 or f.shape[1] == 0)


@_wrap
def neighbourhood(field, hw, stat):
    if hw < 0 or stat == Quantile:
        raise ValueError("arg")
    if _empty2(field):
        return np.zeros((0, 0), np.float32)
    return O.neighbourhood(field, hw, stat)


@_wrap
def neighbourhood_brute_force(field, hw, stat):
    if hw < 0:
        raise ValueError("arg")
    if _empty2(field):
        return np.zeros((0, 0), np.float32)
    return O.neighbourhood_brute_force(field, hw, stat)


@_wrap
def neighbourhood_quantile(field, q, hw):
    if hw < 0:
        raise ValueError("arg")
    if _empty2(field):
        return np.zeros((0, 0), np.float32)
    return O.neighbourhood_quantile(field, q, hw)


@_wrap
def neighbourhood_quantile_fast(field, q, hw, thresholds):
    if hw < 0:
        raise ValueError("arg")
    if _empty2(field):
        return np.zeros((0, 0), np.float32)
    return O.neighbourhood_quantile_fast(field, q, hw, thresholds)


get_neighbourhood_thresholds = _wrap(O.get_neighbourhood_thresholds)
calc_statistic = O.calc_statistic
calc_quantile = _wrap(O.calc_quantile)
calc_even_quantiles = O.calc_even_quantiles


def is_valid(v):
    return bool(O.lib().orc_is_valid(O.C.c_float(v)))


def convert_coordinates(lats, lons, type=Geodetic):
    if np.isscalar(lats):
        x, y, z = O.convert_coordinates([lats], [lons], type)
        return True, float(x[0]), float(y[0]), float(z[0])
    return O.convert_coordinates(lats, lons, type)


def KDTree_calc_distance(lat1, lon1, lat2, lon2, type=Geodetic):
    return float(O.lib().orc_calc_distance(lat1, lon1, lat2, lon2, type))


def KDTree_calc_straight_distance(x0, y0, z0, x1, y1, z1):
    return float(O.lib().orc_calc_straight_distance(x0, y0, z0, x1, y1, z1))


def KDTree_rad2deg(rad):
    return float(np.float32(np.float64(np.float32(rad)) * 180 / np.pi))      # src/api/kdtree.cpp:198-200
