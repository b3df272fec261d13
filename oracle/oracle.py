"""ctypes front-end of the CPU oracle (oracle/gridpp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under gridpp_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

Geodetic, Cartesian = 0, 1
Mean, Min, Median, Max, Quantile, Std, Variance, Sum, Count, RandomChoice = 0, 10, 20, 30, 40, 50, 60, 70, 80, 90

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class OracleError(ValueError):
    pass


class OracleSingular(RuntimeError):
    pass


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "gridpp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_barnes_rho.restype = C.c_float
        L.orc_barnes_rho.argtypes = [C.c_float, C.c_float]
        L.orc_barnes_min_rho.restype = C.c_float
        L.orc_barnes_min_rho.argtypes = [C.c_float, C.c_float]
        L.orc_barnes_localization_distance.restype = C.c_float
        L.orc_barnes_localization_distance.argtypes = [C.c_float, C.c_float]
        L.orc_barnes_corr.restype = C.c_float
        L.orc_barnes_corr.argtypes = [C.c_float] * 14
        L.orc_calc_straight_distance.restype = C.c_float
        L.orc_calc_straight_distance.argtypes = [C.c_float] * 6
        L.orc_calc_distance.restype = C.c_float
        L.orc_calc_distance.argtypes = [C.c_float] * 4 + [C.c_int]
        L.orc_calc_quantile.restype = C.c_float
        L.orc_calc_statistic.restype = C.c_float
        L.orc_interpolate.restype = C.c_float
    return _LIB


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _check(rc):
    if rc == -1:
        raise OracleError("invalid argument")
    if rc == -2:
        raise OracleSingular("singular matrix")
    if rc != 0:
        raise RuntimeError("oracle error %d" % rc)


def convert_coordinates(lats, lons, ctype=Geodetic):
    lats, lons = _f(lats).ravel(), _f(lons).ravel()
    n = lats.size
    x, y, z = (np.empty(n, np.float32) for _ in range(3))
    rc = lib().orc_convert_coordinates(lats.ctypes, lons.ctypes, n, ctype, x.ctypes, y.ctypes, z.ctypes)
    _check(rc)
    return x, y, z


class Pts:
    """Flat point set: lat/lon -> x,y,z (float32) + elev/laf (NaN if missing)."""

    def __init__(self, lats, lons, elevs=None, lafs=None, ctype=Geodetic):
        self.lats, self.lons = _f(lats).ravel(), _f(lons).ravel()
        n = self.lats.size
        self.n = n
        self.ctype = ctype
        self.x, self.y, self.z = convert_coordinates(self.lats, self.lons, ctype)
        self.elevs = _f(elevs).ravel() if elevs is not None and np.size(elevs) == n else np.full(n, np.nan, np.float32)
        self.lafs = _f(lafs).ravel() if lafs is not None and np.size(lafs) == n else np.full(n, np.nan, np.float32)


class Barnes:
    def __init__(self, h, v=0.0, w=0.0, hmax=float("nan")):
        self.h, self.v, self.w = float(h), float(v), float(w)
        self.min_rho = float(lib().orc_barnes_min_rho(h, hmax))

    def localization_distance(self):
        return float(lib().orc_barnes_localization_distance(self.h, self.min_rho))

    def corr(self, p1, p2):
        """p = (x, y, z, elev, laf)"""
        return float(lib().orc_barnes_corr(*[float(t) for t in p1], *[float(t) for t in p2],
                                           self.h, self.v, self.w, self.localization_distance()))


class Struct:
    """Generic scalar structure (Barnes/Cressman/Soar/Toar/Powerlaw/Linear kernels, Multiple mix, CrossValidation)."""
    KINDS = dict(Barnes=0, Cressman=1, Soar=2, Toar=3, Powerlaw=4, Linear=5)

    def __init__(self, kind, h, v=0.0, w=0.0, hmax=float("nan")):
        L = lib()
        L.orc_structure_min_rho.restype = C.c_float
        L.orc_structure_localization_distance.restype = C.c_float
        k = self.KINDS[kind] if isinstance(kind, str) else kind
        self.kh = self.kv = self.kw = k
        self.h, self.v, self.w = float(h), float(v), float(w)
        self.min_rho = float(L.orc_structure_min_rho(C.c_int(k), C.c_float(h), C.c_float(hmax)))
        self.loc = float(L.orc_structure_localization_distance(C.c_int(k), C.c_float(h), C.c_float(self.min_rho)))
        self.cv, self.cv_dist = 0, 0.0

    @staticmethod
    def multiple(sh, sv, sw):
        m = Struct(sh.kh, sh.h)
        m.kh, m.kv, m.kw = sh.kh, sv.kv, sw.kw
        m.h, m.v, m.w, m.min_rho, m.loc = sh.h, sv.v, sw.w, sh.min_rho, sh.loc
        return m

    def cross_validation(self, dist):
        import copy
        c = copy.copy(self)
        c.cv, c.cv_dist = 1, float(dist)
        return c

    def localization_distance(self):
        return self.loc

    def corr(self, p1, p2, background=False):
        L = lib()
        L.orc_corr_generic.restype = C.c_float
        return float(L.orc_corr_generic(C.c_int(self.kh), C.c_int(self.kv), C.c_int(self.kw), C.c_float(self.h), C.c_float(self.v),
                                        C.c_float(self.w), C.c_float(self.loc), C.c_int(self.cv), C.c_float(self.cv_dist),
                                        C.c_int(int(background)), *[C.c_float(float(t)) for t in p1], *[C.c_float(float(t)) for t in p2]))


def oi_full_generic(g, background, bvariance, p, obs, obs_variance, pbackground, bvariance_at_points, st, max_points,
                    allow_extrapolation=True, cell_params=None, obs_params=None):
    """cell_params / obs_params: (h, v, w, R) arrays per background point / per observation for the spatially varying form"""
    background, bvariance = _f(background).ravel(), _f(bvariance).ravel()
    obs, obs_variance, pbackground, bvp = _f(obs), _f(obs_variance), _f(pbackground), _f(bvariance_at_points)
    out = np.empty(g.n, np.float32)
    var = np.empty(g.n, np.float32)
    rc = lib().orc_oi_full_generic(C.c_int(g.n), g.x.ctypes, g.y.ctypes, g.z.ctypes, g.elevs.ctypes, g.lafs.ctypes,
                                   background.ctypes, bvariance.ctypes, C.c_int(p.n), p.x.ctypes, p.y.ctypes, p.z.ctypes,
                                   p.elevs.ctypes, p.lafs.ctypes, obs.ctypes, obs_variance.ctypes, pbackground.ctypes, bvp.ctypes,
                                   C.c_int(st.kh), C.c_int(st.kv), C.c_int(st.kw), C.c_float(st.h), C.c_float(st.v), C.c_float(st.w),
                                   C.c_float(st.loc), C.c_int(st.cv), C.c_float(st.cv_dist), C.c_int(max_points),
                                   C.c_int(1 if allow_extrapolation else 0), out.ctypes, var.ctypes,
                                   *[(_f(a).ctypes if a is not None else None) for a in (cell_params or [None] * 4)],
                                   *[(_f(a).ctypes if a is not None else None) for a in (obs_params or [None] * 4)])
    _check(rc)
    return out, var


def structure_localization(kind, h, min_rho):
    L = lib()
    L.orc_structure_localization_distance.restype = C.c_float
    k = Struct.KINDS[kind] if isinstance(kind, str) else kind
    return float(L.orc_structure_localization_distance(C.c_int(k), C.c_float(h), C.c_float(min_rho)))


def oi_full(g, background, bvariance, p, obs, obs_variance, pbackground, bvariance_at_points,
            st, max_points, allow_extrapolation=True, y0=0, y1=None):
    background, bvariance = _f(background).ravel(), _f(bvariance).ravel()
    obs, obs_variance, pbackground, bvp = _f(obs), _f(obs_variance), _f(pbackground), _f(bvariance_at_points)
    out = np.empty(g.n, np.float32)
    var = np.empty(g.n, np.float32)
    if y1 is None:
        y1 = g.n
    L = lib()
    rc = L.orc_oi_full_range(C.c_int(y0), C.c_int(y1), g.x.ctypes, g.y.ctypes, g.z.ctypes, g.elevs.ctypes, g.lafs.ctypes,
                             background.ctypes, bvariance.ctypes, C.c_int(p.n), p.x.ctypes, p.y.ctypes, p.z.ctypes,
                             p.elevs.ctypes, p.lafs.ctypes, obs.ctypes, obs_variance.ctypes, pbackground.ctypes,
                             bvp.ctypes, C.c_float(st.h), C.c_float(st.v), C.c_float(st.w), C.c_float(st.min_rho),
                             C.c_int(max_points), C.c_int(1 if allow_extrapolation else 0), out.ctypes, var.ctypes)
    _check(rc)
    return out[y0:y1], var[y0:y1]


def oi_baseline(g, background, p, obs, ratios, pbackground, st, max_points, threads=0, cell_list=True, allow_extrapolation=True):
    """The CPU baseline of bench.py: the loop of oi() with a cell-list radius query and OpenMP over the grid points
    (orc_oi_full_omp); bit-identical to oi() by construction, asserted in tests/test_oracle_baseline.py."""
    background = _f(background).ravel()
    obs, ratios, pbackground = _f(obs), _f(ratios), _f(pbackground)
    ones_g, ones_p = np.ones(g.n, np.float32), np.ones(p.n, np.float32)
    out = np.empty(g.n, np.float32)
    var = np.empty(g.n, np.float32)
    rc = lib().orc_oi_full_omp(C.c_int(threads), C.c_int(1 if cell_list else 0), C.c_int(0), C.c_int(g.n), g.x.ctypes, g.y.ctypes,
                               g.z.ctypes, g.elevs.ctypes, g.lafs.ctypes, background.ctypes, ones_g.ctypes, C.c_int(p.n), p.x.ctypes,
                               p.y.ctypes, p.z.ctypes, p.elevs.ctypes, p.lafs.ctypes, obs.ctypes, ratios.ctypes, pbackground.ctypes,
                               ones_p.ctypes, C.c_float(st.h), C.c_float(st.v), C.c_float(st.w), C.c_float(st.min_rho),
                               C.c_int(max_points), C.c_int(1 if allow_extrapolation else 0), out.ctypes, var.ctypes)
    _check(rc)
    return out


def omp_max_threads():
    return int(lib().orc_omp_max_threads())


def set_neighbourhood_threads(n):
    """threads of the loops the reference runs under OpenMP in the neighbourhood family (1 = serial, the default); returns the old value"""
    return int(lib().orc_set_neighbourhood_threads(C.c_int(int(n))))


def oi(g, background, p, obs, ratios, pbackground, st, max_points, allow_extrapolation=True, y0=0, y1=None):
    """optimal_interpolation(Points...) = _full with unit variances (src/api/oi.cpp:123-135)."""
    ones_g = np.ones(g.n, np.float32)
    ones_p = np.ones(p.n, np.float32)
    return oi_full(g, background, ones_g, p, obs, ratios, pbackground, ones_p, st, max_points,
                   allow_extrapolation, y0, y1)[0]


def oi_selection(g, cell, p, obs, pbackground, st, max_points):
    sel = np.empty(max(p.n, 1), np.int32)
    tie = C.c_int(0)
    obs, pbackground = _f(obs), _f(pbackground)
    n = lib().orc_oi_selection(C.c_float(g.x[cell]), C.c_float(g.y[cell]), C.c_float(g.z[cell]),
                               C.c_float(g.elevs[cell]), C.c_float(g.lafs[cell]), C.c_int(p.n),
                               p.x.ctypes, p.y.ctypes, p.z.ctypes, p.elevs.ctypes, p.lafs.ctypes,
                               obs.ctypes, pbackground.ctypes, C.c_float(st.h), C.c_float(st.v), C.c_float(st.w),
                               C.c_float(st.min_rho), C.c_int(max_points), sel.ctypes, C.byref(tie))
    return sel[:n].copy(), bool(tie.value)


def oi_ensi(g, background, p, obs, sigmas, pbackground, st, max_points, allow_extrapolation=True, y0=0, y1=None):
    background = _f(background)
    nY, nE = background.shape
    pbackground = _f(pbackground).reshape(p.n, nE)
    obs, sigmas = _f(obs), _f(sigmas)
    out = np.empty((nY, nE), np.float32)
    if y1 is None:
        y1 = nY
    rc = lib().orc_oi_ensi_range(C.c_int(y0), C.c_int(y1), C.c_int(nY), C.c_int(nE), g.x.ctypes, g.y.ctypes, g.z.ctypes,
                                 g.elevs.ctypes, g.lafs.ctypes, background.ctypes, C.c_int(p.n), p.x.ctypes, p.y.ctypes,
                                 p.z.ctypes, p.elevs.ctypes, p.lafs.ctypes, obs.ctypes, sigmas.ctypes, pbackground.ctypes,
                                 C.c_float(st.h), C.c_float(st.v), C.c_float(st.w), C.c_float(st.min_rho),
                                 C.c_int(max_points), C.c_int(1 if allow_extrapolation else 0), out.ctypes)
    _check(rc)
    return out[y0:y1]


def oi_ensi_generic(g, background, p, obs, sigmas, pbackground, st, max_points, allow_extrapolation=True, cell_params=None):
    """optimal_interpolation_ensi with a Struct (any kernel, Multiple mix, CrossValidation); cell_params = (h, v, w, R) arrays per
    background point for the spatially varying forms (the structure as seen from the grid point, oi_ensi.cpp:213,250)."""
    background = _f(background)
    nY, nE = background.shape
    pbackground = _f(pbackground).reshape(p.n, nE)
    obs, sigmas = _f(obs), _f(sigmas)
    out = np.empty((nY, nE), np.float32)
    cp = [(_f(a) if a is not None else None) for a in (cell_params or [None] * 4)]
    rc = lib().orc_oi_ensi_generic(C.c_int(nY), C.c_int(nE), g.x.ctypes, g.y.ctypes, g.z.ctypes, g.elevs.ctypes, g.lafs.ctypes,
                                   background.ctypes, C.c_int(p.n), p.x.ctypes, p.y.ctypes, p.z.ctypes, p.elevs.ctypes, p.lafs.ctypes,
                                   obs.ctypes, sigmas.ctypes, pbackground.ctypes,
                                   C.c_int(st.kh), C.c_int(st.kv), C.c_int(st.kw), C.c_float(st.h), C.c_float(st.v), C.c_float(st.w),
                                   C.c_float(st.loc), C.c_int(st.cv), C.c_float(st.cv_dist), C.c_int(max_points),
                                   C.c_int(1 if allow_extrapolation else 0), out.ctypes,
                                   *[(a.ctypes if a is not None else None) for a in cp])
    _check(rc)
    return out


def oi_ensi_multi(variant, g, bratios, background, background_corr, p, pobs, pratios, pbackground, pbackground_corr, st, max_points,
                  allow_extrapolation=True):
    """optimal_interpolation_ensi_multi_{ebe, ebesc, utem}: variant 'ebe' | 'ebesc' | 'utem' (background_corr / pbackground_corr
    are ignored by ebesc)."""
    vid = {"ebe": 1, "ebesc": 2, "utem": 3}[variant]
    background = _f(background)
    nY, nE = background.shape
    bc = _f(background_corr).reshape(nY, nE) if background_corr is not None else background
    pb = _f(pbackground).reshape(p.n, nE)
    pbc = _f(pbackground_corr).reshape(p.n, nE) if pbackground_corr is not None else pb
    pobs = _f(pobs).reshape(p.n) if vid == 3 else _f(pobs).reshape(p.n, nE)
    bratios, pratios = _f(bratios).ravel(), _f(pratios).ravel()
    out = np.empty((nY, nE), np.float32)
    rc = lib().orc_oi_ensi_multi(C.c_int(vid), C.c_int(nY), C.c_int(nE), g.x.ctypes, g.y.ctypes, g.z.ctypes, g.elevs.ctypes, g.lafs.ctypes,
                                 bratios.ctypes, background.ctypes, bc.ctypes, C.c_int(p.n), p.x.ctypes, p.y.ctypes, p.z.ctypes,
                                 p.elevs.ctypes, p.lafs.ctypes, pobs.ctypes, pratios.ctypes, pb.ctypes, pbc.ctypes,
                                 C.c_float(st.h), C.c_float(st.v), C.c_float(st.w), C.c_float(st.min_rho),
                                 C.c_int(max_points), C.c_int(1 if allow_extrapolation else 0), out.ctypes)
    _check(rc)
    return out


def get_neighbours(p, qlat, qlon, radius, include_match=True):
    qx, qy, qz = convert_coordinates([qlat], [qlon], p.ctype)
    out = np.empty(max(p.n, 1), np.int32)
    n = lib().orc_get_neighbours(p.x.ctypes, p.y.ctypes, p.z.ctypes, C.c_int(p.n), C.c_float(qx[0]), C.c_float(qy[0]),
                                 C.c_float(qz[0]), C.c_float(radius), C.c_int(int(include_match)), out.ctypes)
    return out[:n].copy()


def nearest_neighbour(p, qlat, qlon, include_match=True):
    qx, qy, qz = convert_coordinates([qlat], [qlon], p.ctype)
    return int(lib().orc_nearest_neighbour(p.x.ctypes, p.y.ctypes, p.z.ctypes, C.c_int(p.n), C.c_float(qx[0]),
                                           C.c_float(qy[0]), C.c_float(qz[0]), C.c_int(int(include_match))))


def nearest_indices(g, q):
    out = np.empty(q.n, np.float32)
    idx = np.empty(q.n, np.int32)
    vals = np.zeros(max(g.n, 1), np.float32)
    lib().orc_nearest(g.x.ctypes, g.y.ctypes, g.z.ctypes, C.c_int(g.n), vals.ctypes, q.x.ctypes, q.y.ctypes, q.z.ctypes,
                      C.c_int(q.n), out.ctypes, idx.ctypes)
    return idx


def nearest(g, q, values):
    values = _f(values).ravel()
    out = np.empty(q.n, np.float32)
    lib().orc_nearest(g.x.ctypes, g.y.ctypes, g.z.ctypes, C.c_int(g.n), values.ctypes, q.x.ctypes, q.y.ctypes, q.z.ctypes,
                      C.c_int(q.n), out.ctypes, None)
    return out


def count(p, q, radius):
    """count(input set p, output locations q, radius) (src/api/count.cpp)"""
    out = np.empty(q.n, np.float32)
    lib().orc_count(p.x.ctypes, p.y.ctypes, p.z.ctypes, C.c_int(p.n), q.x.ctypes, q.y.ctypes, q.z.ctypes, C.c_int(q.n),
                    C.c_float(radius), out.ctypes)
    return out


def gridding(q, p, values, radius, min_num, statistic):
    """gridding(output locations q, input points p, values, ...) (src/api/gridding.cpp:6-63)"""
    values = _f(values).ravel()
    if values.size != p.n:
        raise OracleError("Points size is not the same as values")
    out = np.empty(q.n, np.float32)
    v = values if values.size else np.zeros(1, np.float32)
    _check(lib().orc_gridding(p.x.ctypes, p.y.ctypes, p.z.ctypes, v.ctypes, C.c_int(p.n), q.x.ctypes, q.y.ctypes, q.z.ctypes,
                              C.c_int(q.n), C.c_float(radius), C.c_int(min_num), C.c_int(statistic), out.ctypes))
    return out


def gridding_nearest(q, p, values, min_num, statistic):
    """gridding_nearest(output locations q, input points p, values, ...) (src/api/gridding.cpp:65-131)"""
    values = _f(values).ravel()
    if values.size != p.n:
        raise OracleError("Points size is not the same as values")
    out = np.empty(q.n, np.float32)
    v = values if values.size else np.zeros(1, np.float32)
    _check(lib().orc_gridding_nearest(q.x.ctypes, q.y.ctypes, q.z.ctypes, C.c_int(q.n), p.x.ctypes, p.y.ctypes, p.z.ctypes,
                                      v.ctypes, C.c_int(p.n), C.c_int(min_num), C.c_int(statistic), out.ctypes))
    return out


def fill(g, input, p, radii, value, outside):
    input = _f(input)
    radii = _f(radii).ravel()
    out = np.empty(input.size, np.float32)
    r = radii if radii.size else np.zeros(1, np.float32)
    _check(lib().orc_fill(g.x.ctypes, g.y.ctypes, g.z.ctypes, C.c_int(g.n), np.ascontiguousarray(input).ctypes, p.x.ctypes, p.y.ctypes,
                          p.z.ctypes, r.ctypes, C.c_int(p.n), C.c_float(value), C.c_int(int(bool(outside))), out.ctypes))
    return out.reshape(input.shape)


def fill_missing(values):
    values = np.ascontiguousarray(_f(values))
    out = np.empty_like(values)
    _check(lib().orc_fill_missing(values.ctypes, C.c_int(values.shape[0]), C.c_int(values.shape[1]), out.ctypes))
    return out


def doping_square(g, shape, background, p, obs, halfwidth, max_elev_diff):
    background = np.ascontiguousarray(_f(background))
    hw = np.ascontiguousarray(np.asarray(halfwidth, np.int32).ravel())
    obs = _f(obs).ravel()
    out = np.empty(background.size, np.float32)
    _check(lib().orc_doping_square(g.x.ctypes, g.y.ctypes, g.z.ctypes, g.elevs.ctypes, C.c_int(shape[0]), C.c_int(shape[1]),
                                   background.ctypes, p.x.ctypes, p.y.ctypes, p.z.ctypes, p.elevs.ctypes, obs.ctypes, hw.ctypes,
                                   C.c_int(p.n), C.c_float(max_elev_diff), out.ctypes))
    return out.reshape(background.shape)


def doping_circle(g, background, p, obs, radii, max_elev_diff):
    background = np.ascontiguousarray(_f(background))
    obs, radii = _f(obs).ravel(), _f(radii).ravel()
    out = np.empty(background.size, np.float32)
    _check(lib().orc_doping_circle(g.x.ctypes, g.y.ctypes, g.z.ctypes, g.elevs.ctypes, C.c_int(g.n), background.ctypes, p.x.ctypes,
                                   p.y.ctypes, p.z.ctypes, p.elevs.ctypes, obs.ctypes, radii.ctypes, C.c_int(p.n),
                                   C.c_float(max_elev_diff), out.ctypes))
    return out.reshape(background.shape)


def neighbourhood_search(array, search, halfwidth, tmin, tmax, delta, apply=None):
    array, search = np.ascontiguousarray(_f(array)), np.ascontiguousarray(_f(search))
    out = np.empty_like(array)
    ap = None if apply is None else np.ascontiguousarray(np.asarray(apply).astype(np.int32))
    _check(lib().orc_neighbourhood_search(array.ctypes, search.ctypes, C.c_int(array.shape[0]), C.c_int(array.shape[1]), C.c_int(halfwidth),
                                          C.c_float(tmin), C.c_float(tmax), C.c_float(delta), None if ap is None else ap.ctypes, out.ctypes))
    return out


def calc_gradient(base, values, gradient_type, halfwidth, num_min, min_range, default_gradient):
    base, values = np.ascontiguousarray(_f(base)), np.ascontiguousarray(_f(values))
    out = np.empty_like(base)
    _check(lib().orc_calc_gradient(base.ctypes, values.ctypes, C.c_int(base.shape[0]), C.c_int(base.shape[1] if base.ndim == 2 else 0),
                                   C.c_int(gradient_type), C.c_int(halfwidth), C.c_int(num_min), C.c_float(min_range),
                                   C.c_float(default_gradient), out.ctypes))
    return out


def staticcorr_points(g, p, st, max_points):
    """g: the points, p: the knots, st: Struct (src/api/corr_points.cpp:26-131)"""
    out = np.empty((g.n, p.n), np.float32)
    _check(lib().orc_staticcorr_points(C.c_int(g.n), g.x.ctypes, g.y.ctypes, g.z.ctypes, g.elevs.ctypes, g.lafs.ctypes, C.c_int(p.n),
                                       p.x.ctypes, p.y.ctypes, p.z.ctypes, p.elevs.ctypes, p.lafs.ctypes, C.c_int(st.kh), C.c_int(st.kv),
                                       C.c_int(st.kw), C.c_float(st.h), C.c_float(st.v), C.c_float(st.w), C.c_float(st.loc), C.c_int(st.cv),
                                       C.c_float(st.cv_dist), C.c_int(max_points), out.ctypes))
    return out


def distance(p, q, num, query_first):
    """distance(input set p, output locations q, num) (src/api/distance.cpp)"""
    out = np.empty(q.n, np.float32)
    lib().orc_distance(p.x.ctypes, p.y.ctypes, p.z.ctypes, p.lats.ctypes, p.lons.ctypes, C.c_int(p.n), q.x.ctypes, q.y.ctypes, q.z.ctypes,
                       q.lats.ctypes, q.lons.ctypes, C.c_int(q.n), C.c_int(num), C.c_int(p.ctype), C.c_int(int(query_first)), out.ctypes)
    return out


class OracleDistorted(RuntimeError):
    """bilinear: s / t outside [0, 1] (the reference throws std::runtime_error, src/api/bilinear.cpp:309-313)"""


def point_in_rectangle(A, B, C_, D, m):
    """A..D, m = (lat, lon) pairs (src/api/util.cpp:571-582)"""
    f = lib().orc_point_in_rectangle
    f.argtypes = [C.c_float] * 10
    return bool(f(A[0], A[1], B[0], B[1], C_[0], C_[1], D[0], D[1], m[0], m[1]))


def get_box(g, shape, lat, lon):
    """Grid::get_box (src/api/grid.cpp:149-229) -> [inside, Y1, X1, Y2, X2]"""
    nY, nX = shape
    if g.n == 0:
        return [False, -1, -1, -1, -1]
    nn = nearest_neighbour(g, lat, lon, True)
    box = (C.c_int * 4)()
    f = lib().orc_get_box
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float] + [C.POINTER(C.c_int)] * 4
    inside = f(g.lats.ctypes.data, g.lons.ctypes.data, nY, nX, nn, lat, lon, *[C.cast(C.byref(box, 4 * k), C.POINTER(C.c_int)) for k in range(4)])
    return [bool(inside)] + [int(b) for b in box]


def bilinear(g, shape, q, values):
    """g: Pts of a (Y, X) grid, q: Pts of the output locations, values (Y, X) or (T, Y, X) -> (nq,) or (T, nq)"""
    nY, nX = shape
    values = _f(values)
    three_d = values.ndim == 3
    nT = values.shape[0] if three_d else 1
    out = np.empty((nT, q.n), np.float32)
    bad = np.zeros(2, np.float32)
    v = np.ascontiguousarray(values).ravel()
    if v.size == 0:
        v = np.zeros(1, np.float32)
    rc = lib().orc_bilinear(g.lats.ctypes, g.lons.ctypes, g.x.ctypes, g.y.ctypes, g.z.ctypes, C.c_int(nY), C.c_int(nX),
                            v.ctypes, C.c_int(nT), q.lats.ctypes, q.lons.ctypes, q.x.ctypes, q.y.ctypes, q.z.ctypes,
                            C.c_int(q.n), out.ctypes, bad[0:].ctypes, bad[1:].ctypes)
    if rc == -2:
        raise OracleDistorted("Problem with bilinear interpolation. s=%g and t=%g" % (bad[0], bad[1]))
    _check(rc)
    return out if three_d else out[0]


def calc_statistic(a, stat):
    a = _f(a).ravel()
    return float(lib().orc_calc_statistic(a.ctypes, C.c_int(a.size), C.c_int(stat)))


def calc_quantile(a, q):
    a = _f(a).ravel()
    err = C.c_int(0)
    r = float(lib().orc_calc_quantile(a.ctypes, C.c_int(a.size), C.c_float(q), C.byref(err)))
    _check(err.value)
    return r


def interpolate(x, ix, iy):
    ix, iy = _f(ix), _f(iy)
    return float(lib().orc_interpolate(C.c_float(x), ix.ctypes, iy.ctypes, C.c_int(ix.size)))


def calc_even_quantiles(values, num):
    values = _f(values).ravel()
    out = np.empty(max(num, values.size, 1), np.float32)
    n = lib().orc_calc_even_quantiles(values.ctypes, C.c_int(values.size), C.c_int(num), out.ctypes)
    return out[:n].copy()


def get_neighbourhood_thresholds(field, num):
    field = _f(field).ravel()
    out = np.empty(max(num, 1), np.float32)
    n = lib().orc_get_neighbourhood_thresholds(field.ctypes, C.c_long(field.size), C.c_int(num), out.ctypes)
    if n < 0:
        _check(n)
    return out[:n].copy()


def _yx(field):
    field = _f(field)
    if field.ndim == 2:
        return field, field.shape[0], field.shape[1], 1, 0
    return field, field.shape[0], field.shape[1], field.shape[2], 1


def neighbourhood(field, hw, stat):
    field, Y, X, E, is3d = _yx(field)
    out = np.empty((Y, X), np.float32)
    if is3d:
        rc = lib().orc_neighbourhood3(field.ctypes, C.c_int(Y), C.c_int(X), C.c_int(E), C.c_int(hw), C.c_int(stat), out.ctypes)
    else:
        rc = lib().orc_neighbourhood(field.ctypes, C.c_int(Y), C.c_int(X), C.c_int(hw), C.c_int(stat), out.ctypes)
    _check(rc)
    return out


def neighbourhood_brute_force(field, hw, stat):
    field, Y, X, E, is3d = _yx(field)
    out = np.empty((Y, X), np.float32)
    _check(lib().orc_neighbourhood_brute_force(field.ctypes, C.c_int(Y), C.c_int(X), C.c_int(E), C.c_int(hw), C.c_int(stat), out.ctypes))
    return out


def neighbourhood_quantile(field, q, hw):
    field, Y, X, E, is3d = _yx(field)
    out = np.empty((Y, X), np.float32)
    _check(lib().orc_neighbourhood_quantile(field.ctypes, C.c_int(Y), C.c_int(X), C.c_int(E), C.c_float(q), C.c_int(hw), out.ctypes))
    return out


def neighbourhood_quantile_fast(field, q, hw, thresholds):
    field, Y, X, E, is3d = _yx(field)
    q = _f(q).ravel()
    thresholds = _f(thresholds).ravel()
    out = np.empty((Y, X), np.float32)
    _check(lib().orc_neighbourhood_quantile_fast(field.ctypes, C.c_int(Y), C.c_int(X), C.c_int(E), C.c_int(is3d), q.ctypes,
                                                 C.c_int(q.size), C.c_int(hw), thresholds.ctypes, C.c_int(thresholds.size),
                                                 out.ctypes))
    return out
