"""gridpp_amd -- the python surface of the MI355X-native gridpp hot path.

Mirrors the SWIG module of the reference (swig/gridpp.i %include "gridpp.h"):
same names, argument order, defaults, output dtypes (np.float32 / np.int32) and
exception mapping (ValueError for invalid arguments, RuntimeError otherwise), for
the optimal-interpolation + neighbourhood path only.  Use as

    import gridpp_amd as gridpp

All compute goes through libgridpp_hip.so (include/gridpp_hip.h).  Field
arguments may be anything numpy can convert (host path: staged through HBM by the
library) or torch CUDA tensors (device path: the kernels read/write the tensors'
HBM directly and a torch tensor is returned).
"""
import ctypes as C
import os
import weakref

import numpy as np

from . import _capi
from ._capi import check, lib

__version__ = "0.8.0.dev1+mi355x.r1"

# include/gridpp.h:120-123, :88-100
Geodetic, Cartesian = 0, 1
Mean, Min, Median, Max, Quantile, Std, Variance, Sum, Count, RandomChoice, Unknown = 0, 10, 20, 30, 40, 50, 60, 70, 80, 90, -1
MV = np.nan
radius_earth = 6.378137e6


def version():
    return lib().gpp_version().decode()


def set_device(index):
    """One process per GPU: call with LOCAL_RANK before anything else."""
    check(lib().gpp_set_device(int(index)))


def device_count():
    n = C.c_int(0)
    check(lib().gpp_device_count(C.byref(n)))
    return n.value


def synchronize():
    check(lib().gpp_synchronize())


def is_valid(value):
    """src/api/util.cpp:16-18"""
    v = np.float32(value)
    return bool(not np.isnan(v) and not np.isinf(v))


_omp_threads = 1


def get_statistic(name):
    """gridpp::get_statistic (include/gridpp.h:1410, src/api/gridpp.cpp:11-43): the Statistic of a name.  The reference's table has no
    "variance": that name, like any other it does not know, gives Unknown."""
    return {"mean": Mean, "min": Min, "max": Max, "median": Median, "quantile": Quantile, "std": Std, "sum": Sum, "count": Count,
            "randomchoice": RandomChoice}.get(name, Unknown)


def set_omp_threads(num):   # src/api/gridpp.cpp:184-207 -- meaningless on the GPU path, kept for drop-in
    global _omp_threads
    _omp_threads = int(num)


def get_omp_threads():
    return _omp_threads


# ---- messages (include/gridpp.h:1394-1430, src/api/gridpp.cpp:70-76, src/api/util.cpp:226-252) ------------
_debug_level = 0


def set_debug_level(level):
    global _debug_level
    _debug_level = int(level)


def get_debug_level():
    return _debug_level


def debug(string):
    print(string, flush=True)


def warning(string):
    print("Warning: " + str(string), flush=True)


def error(string):
    print("Error: " + str(string), flush=True)
    raise RuntimeError(string)


def future_deprecation_warning(function, other=""):
    print("Future deprecation warning: %s will be deprecated%s" % (function, (", use %s instead." % other) if other else "."), flush=True)


def clock():
    """seconds since the epoch (src/api/util.cpp:254-260)"""
    import time
    return time.time()


# ---- argument conversion (the SWIG typemaps of swig/vector.i) ---------------------------------
def _is_dev(a):
    return hasattr(a, "data_ptr") and getattr(a, "is_cuda", False)


def _wants_f64(*fields):
    """The large field arrays of a call are float64 numpy arrays: hand every input over as float64 (GPP_HOST_F64, cast on the
    device) instead of paying numpy's astype on them."""
    big = [a for a in fields if isinstance(a, np.ndarray) and a.size >= (1 << 20)]
    return bool(big) and all(a.dtype == np.float64 for a in big)


def _vec(a, ndim, name="array", dtype=np.float32):
    """Any dtype -> C-contiguous float32 (float64 in a GPP_HOST_F64 call); wrong ndim raises like the typemap (swig/vector.i:39-41)."""
    if _is_dev(a):
        import torch
        if a.dim() != ndim:
            raise RuntimeError("%s must have %d dimensions" % (name, ndim))
        return a.contiguous().to(torch.float32)
    arr = np.ascontiguousarray(np.asarray(a), dtype=dtype)
    if arr.ndim != ndim:
        if arr.size == 0 and arr.ndim <= ndim:   # e.g. [] or [[]]
            return arr.reshape((0,) * ndim)
        raise RuntimeError("%s must have %d dimensions, got %d" % (name, ndim, arr.ndim))
    return arr


def _ptr(a):
    if a is None:
        return None
    if _is_dev(a):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(a.ctypes.data)


def _shape(a):
    return tuple(a.shape)


class _PinnedPool:
    """Result arrays of 1 MiB and more live in page-locked host memory (gpp_host_alloc): the device-to-host copy is then one
    DMA transfer.  Buffers return to a free list when their numpy array dies (up to 2 GiB are kept for reuse)."""
    MIN_BYTES, KEEP_BYTES = 1 << 20, 2 << 30

    def __init__(self):
        self.free, self.kept = {}, 0

    def take(self, nbytes):
        lst = self.free.get(nbytes)
        if lst:
            self.kept -= nbytes
            return lst.pop()
        ptr = C.c_void_p()
        if lib().gpp_host_alloc(nbytes, C.byref(ptr)) != _capi.GPP_OK or not ptr.value:
            return None
        return ptr.value

    def give(self, ptr, nbytes):
        if self.kept + nbytes <= self.KEEP_BYTES:
            self.free.setdefault(nbytes, []).append(ptr)
            self.kept += nbytes
        else:
            lib().gpp_host_free(C.c_void_p(ptr))


_pinned = _PinnedPool()


def _host_empty(shape):
    n = int(np.prod(shape))
    if 4 * n < _PinnedPool.MIN_BYTES or os.environ.get("GPP_PAGEABLE_RESULTS"):
        return np.empty(shape, dtype=np.float32)
    ptr = _pinned.take(4 * n)
    if ptr is None:
        return np.empty(shape, dtype=np.float32)
    buf = (C.c_float * n).from_address(ptr)
    weakref.finalize(buf, _pinned.give, ptr, 4 * n)
    return np.frombuffer(buf, dtype=np.float32).reshape(shape)     # the array keeps `buf` alive


def _empty_like_field(shape, like):
    if _is_dev(like):
        import torch
        return torch.empty(shape, dtype=torch.float32, device=like.device)
    return _host_empty(shape)


def _mem(*arrays):
    dev = [_is_dev(a) for a in arrays if a is not None]
    if any(dev) and not all(dev):
        raise ValueError("either all field arguments are torch CUDA tensors or none is")
    return _capi.MEM_DEVICE if dev and all(dev) else _capi.MEM_HOST


# ---- Point / KDTree / Points / Grid -----------------------------------------------------------
class Point:
    """include/gridpp.h:1713-1743, src/api/point.cpp:5-26"""

    def __init__(self, lat, lon, elev=MV, laf=MV, type=Geodetic, x=None, y=None, z=None):
        self.lat, self.lon, self.elev, self.laf, self.type = float(lat), float(lon), float(elev), float(laf), type
        if x is not None:
            self.x, self.y, self.z = float(x), float(y), float(z)
        elif type == Geodetic:
            xs, ys, zs = convert_coordinates([lat], [lon], type)
            self.x, self.y, self.z = float(xs[0]), float(ys[0]), float(zs[0])
        else:   # point.cpp:18-21 (x = lat, y = lon)
            self.x, self.y, self.z = float(np.float32(lat)), float(np.float32(lon)), 0.0

    def _seven(self):
        return (C.c_float * 7)(self.x, self.y, self.z, self.elev, self.laf, self.lat, self.lon)


def convert_coordinates(lats, lons, type=Geodetic):
    """src/api/util.cpp:583-615; the scalar overload (:617-633) returns (ok, x, y, z)"""
    if np.isscalar(lats) and np.isscalar(lons):
        x, y, z = convert_coordinates([lats], [lons], type)
        return True, float(x[0]), float(y[0]), float(z[0])
    lats, lons = _vec(lats, 1, "lats"), _vec(lons, 1, "lons")
    if lats.size != lons.size:
        raise ValueError("lats and lons must have the same size")
    n = lats.size
    x, y, z = (np.empty(n, np.float32) for _ in range(3))
    check(lib().gpp_convert_coordinates(_ptr(lats), _ptr(lons), n, type, _ptr(x), _ptr(y), _ptr(z)))
    return x, y, z


def _is_big_f64(*arrays):
    """float64 numpy coordinate arrays of a large set: handed to the C-ABI as they are and cast to float32 on the device
    (numpy's astype on 16 M doubles takes longer than building the whole Grid)"""
    return all(isinstance(a, np.ndarray) and a.dtype == np.float64 for a in arrays) and arrays[0].size >= (1 << 16)


def _vec64(a, ndim, name):
    arr = np.ascontiguousarray(np.asarray(a), dtype=np.float64)
    if arr.ndim != ndim:
        if arr.size == 0 and arr.ndim <= ndim:
            return arr.reshape((0,) * ndim)
        raise RuntimeError("%s must have %d dimensions, got %d" % (name, ndim, arr.ndim))
    return arr


class _PointSet:
    """Owns a gpp_points handle (x/y/z resident in HBM)."""
    _h = None

    def __del__(self):
        try:
            if self._h is not None:
                lib().gpp_points_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _field(self, k):
        out = np.empty(self._n, np.float32)
        check(lib().gpp_points_get(self._h, k, _ptr(out)))
        return out

    def get_coordinate_type(self):
        return self._type

    # KDTree queries (src/api/kdtree.cpp:18-106)
    def _neighbours(self, lat, lon, radius, include_match=True, want_dist=False):
        cap = max(self._n, 1)
        idx = np.empty(cap, np.int32)
        dist = np.empty(cap, np.float32) if want_dist else None
        cnt = C.c_int(0)
        check(lib().gpp_points_get_neighbours(self._h, float(lat), float(lon), float(radius), int(bool(include_match)),
                                              _ptr(idx), _ptr(dist), cap, C.byref(cnt)))
        if want_dist:
            return idx[:cnt.value].copy(), dist[:cnt.value].copy()
        return idx[:cnt.value].copy()

    def _closest(self, lat, lon, num, include_match=True):
        idx = np.empty(max(int(num), 1), np.int32)
        cnt = C.c_int(0)
        check(lib().gpp_points_get_closest_neighbours(self._h, float(lat), float(lon), int(num), int(bool(include_match)),
                                                      _ptr(idx), C.byref(cnt)))
        return idx[:cnt.value].copy()

    def _nearest_flat(self, lats, lons, include_match=True):
        lats, lons = _vec(lats, 1), _vec(lons, 1)
        out = np.empty(lats.size, np.int32)
        check(lib().gpp_points_nearest_neighbour(self._h, _ptr(lats), _ptr(lons), lats.size, int(bool(include_match)), _ptr(out)))
        return out


class Points(_PointSet):
    """include/gridpp.h:1876-1968, src/api/points.cpp"""

    def __init__(self, lats=(), lons=(), elevs=(), lafs=(), type=Geodetic):
        f64 = _is_big_f64(lats, lons)
        vec = _vec64 if f64 else _vec
        lats, lons = vec(lats, 1, "lats"), vec(lons, 1, "lons")
        elevs, lafs = vec(elevs, 1, "elevs"), vec(lafs, 1, "lafs")
        n = lats.size
        if lons.size != n:
            raise ValueError("Cannot create points with unequal lat and lon sizes")
        if elevs.size not in (0, n):
            raise ValueError("'elevs' must either be size 0 or the same size at lats/lons")
        if lafs.size not in (0, n):
            raise ValueError("'lafs' must either be size 0 or the same size at lats/lons")
        h = C.c_void_p()
        create = lib().gpp_points_create_f64 if f64 else lib().gpp_points_create
        check(create(_ptr(lats), _ptr(lons), _ptr(elevs) if elevs.size == n and n else None,
                     _ptr(lafs) if lafs.size == n and n else None, n, type, C.byref(h)))
        self._h, self._n, self._type = h, n, type

    def size(self):
        return self._n

    def get_lats(self):
        return self._field(0)

    def get_lons(self):
        return self._field(1)

    def get_elevs(self):
        return self._field(2)

    def get_lafs(self):
        return self._field(3)

    def get_neighbours(self, lat, lon, radius, include_match=True):
        return self._neighbours(lat, lon, radius, include_match)

    def get_neighbours_with_distance(self, lat, lon, radius, include_match=True):
        return self._neighbours(lat, lon, radius, include_match, True)

    def get_num_neighbours(self, lat, lon, radius, include_match=True):
        return len(self._neighbours(lat, lon, radius, include_match))

    def get_nearest_neighbour(self, lat, lon, include_match=True):
        return int(self._nearest_flat([lat], [lon], include_match)[0])

    def get_closest_neighbours(self, lat, lon, num, include_match=True):
        return self._closest(lat, lon, num, include_match)

    def get_point(self, index):   # src/api/points.cpp:128-130
        f = [self._field(k)[index] for k in range(7)]
        return Point(f[0], f[1], f[2], f[3], self._type, f[4], f[5], f[6])

    def get_in_domain_indices(self, grid):   # src/api/points.cpp:77-92: the points some grid box encloses (Grid::get_box)
        if self._n == 0 or grid._n == 0:
            return np.zeros(0, np.int32)
        lats, lons = np.ascontiguousarray(self.get_lats(), np.float32), np.ascontiguousarray(self.get_lons(), np.float32)
        inside, boxes = np.zeros(self._n, np.int32), np.zeros(4 * self._n, np.int32)
        check(lib().gpp_grid_get_box(grid._h, _ptr(lats), _ptr(lons), self._n, _ptr(inside), _ptr(boxes)))
        return np.nonzero(inside)[0].astype(np.int32)

    def get_in_domain(self, grid):           # points.cpp:93-109 (the new set is built with the default coordinate type)
        idx = self.get_in_domain_indices(grid)
        return Points(self.get_lats()[idx], self.get_lons()[idx], self.get_elevs()[idx], self.get_lafs()[idx])

    def subset(self, indices):
        indices = np.asarray(indices, np.int64)
        if indices.size and indices.max() >= self._n:
            raise ValueError("Index exceeds number of points")
        return Points(self.get_lats()[indices], self.get_lons()[indices], self.get_elevs()[indices], self.get_lafs()[indices])


class KDTree(Points):
    """include/gridpp.h:1746-1873 (a Points without elev/laf)"""

    def __init__(self, lats=(), lons=(), type=Geodetic):
        Points.__init__(self, lats, lons, (), (), type)

    # static distance helpers (src/api/kdtree.cpp:107-200); host scalar math like the reference's
    @staticmethod
    def calc_straight_distance(*args):
        """(p1, p2) or (x0, y0, z0, x1, y1, z1): 3-D chord length in float32 (kdtree.cpp:189-194)"""
        if len(args) == 2:
            p1, p2 = args
            args = (p1.x, p1.y, p1.z, p2.x, p2.y, p2.z)
        f = np.float32
        x0, y0, z0, x1, y1, z1 = (f(a) for a in args)
        return float(np.sqrt((x0 - x1) * (x0 - x1) + (y0 - y1) * (y0 - y1) + (z0 - z1) * (z0 - z1), dtype=np.float32))

    @staticmethod
    def deg2rad(deg):
        return float(np.float32(np.float64(np.float32(deg)) * np.pi / 180))

    @staticmethod
    def rad2deg(rad):
        return float(np.float32(np.float64(np.float32(rad)) * 180 / np.pi))

    @staticmethod
    def calc_distance(*args):
        """(p1, p2) or (lat1, lon1, lat2, lon2, type=Geodetic): great-circle distance (kdtree.cpp:107-136,182-187)"""
        if len(args) == 2:
            p1, p2 = args
            if p1.type != p2.type:
                raise RuntimeError("Coordinate types must be the same")
            args = (p1.lat, p1.lon, p2.lat, p2.lon, p1.type)
        lat1, lon1, lat2, lon2 = (np.float32(a) for a in args[:4])
        ctype = args[4] if len(args) > 4 else Geodetic
        if ctype == Cartesian:
            dx, dy = lon1 - lon2, lat1 - lat2
            return float(np.sqrt(dx * dx + dy * dy, dtype=np.float32))
        if lat1 == lat2 and lon1 == lon2:
            return 0.0
        r = [np.float64(np.float32(np.float64(v) * np.pi / 180)) for v in (lat1, lat2, lon1, lon2)]   # deg2rad returns float
        lat1r, lat2r, lon1r, lon2r = r
        ratio = (np.cos(lat1r) * np.cos(lon1r) * np.cos(lat2r) * np.cos(lon2r) + np.cos(lat1r) * np.sin(lon1r) * np.cos(lat2r) * np.sin(lon2r)
                 + np.sin(lat1r) * np.sin(lat2r))
        return float(np.float32(np.arccos(ratio) * 6.378137e6))

    @staticmethod
    def calc_distance_fast(*args):
        """(p1, p2) or (lat1, lon1, lat2, lon2, type=Geodetic): equirectangular approximation (kdtree.cpp:137-181)"""
        if len(args) == 2:
            p1, p2 = args
            args = (p1.lat, p1.lon, p2.lat, p2.lon, p1.type)
        lat1, lon1, lat2, lon2 = (np.float32(a) for a in args[:4])
        ctype = args[4] if len(args) > 4 else Geodetic
        if ctype == Cartesian:
            dx, dy = lon1 - lon2, lat1 - lat2
            return float(np.sqrt(dx * dx + dy * dy, dtype=np.float32))
        lat1r, lat2r, lon1r, lon2r = [np.float64(np.float32(np.float64(v) * np.pi / 180)) for v in (lat1, lat2, lon1, lon2)]
        dlon = np.fmod(np.abs(lon1r - lon2r), 2 * np.pi)
        if dlon > np.pi:
            dlon = 2 * np.pi - dlon
        max_lat = lat2r if np.abs(lat2r) > np.abs(lat1r) else lat1r
        dx2 = np.float32(np.cos(max_lat) ** 2 * dlon * dlon)
        dy2 = np.float32((lat1r - lat2r) * (lat1r - lat2r))
        return float(np.float32(6.378137e6 * np.sqrt(np.float64(dx2 + dy2))))


KDTree_calc_distance = KDTree.calc_distance
KDTree_calc_distance_fast = KDTree.calc_distance_fast
KDTree_calc_straight_distance = KDTree.calc_straight_distance
KDTree_deg2rad = KDTree.deg2rad
KDTree_rad2deg = KDTree.rad2deg


class Grid(_PointSet):
    """include/gridpp.h:1971-2060, src/api/grid.cpp"""

    def __init__(self, lats=((),), lons=((),), elevs=((),), lafs=((),), type=Geodetic):
        f64 = _is_big_f64(lats, lons)
        vec = _vec64 if f64 else _vec
        lats, lons = vec(lats, 2, "lats"), vec(lons, 2, "lons")
        elevs, lafs = vec(elevs, 2, "elevs"), vec(lafs, 2, "lafs")
        if lats.shape != lons.shape:
            raise ValueError("lats and lons must have the same shape")
        ny, nx = lats.shape
        if ny * nx == 0:
            ny = nx = 0
        e = elevs if elevs.shape == lats.shape and elevs.size else None   # grid.cpp:41-54
        l = lafs if lafs.shape == lats.shape and lafs.size else None
        h = C.c_void_p()
        create = lib().gpp_grid_create_f64 if f64 else lib().gpp_grid_create
        check(create(_ptr(lats), _ptr(lons), _ptr(e), _ptr(l), ny, nx, type, C.byref(h)))
        self._h, self._n, self._ny, self._nx, self._type = h, ny * nx, ny, nx, type

    def size(self):
        return [self._ny, self._nx]

    def _f2(self, k):
        return self._field(k).reshape(self._ny, self._nx)

    def get_lats(self):
        return self._f2(0)

    def get_lons(self):
        return self._f2(1)

    def get_elevs(self):
        return self._f2(2)

    def get_lafs(self):
        return self._f2(3)

    def get_nearest_neighbour(self, lat, lon, include_match=True):   # grid.cpp:76-82,108-114
        if self._n == 0:
            return []
        i = int(self._nearest_flat([lat], [lon], include_match)[0])
        return [i // self._nx, i % self._nx]

    def get_neighbours(self, lat, lon, radius, include_match=True):
        idx = self._neighbours(lat, lon, radius, include_match)
        return np.stack([idx // self._nx, idx % self._nx], axis=1).astype(np.int32) if idx.size else np.zeros((0, 2), np.int32)

    def get_closest_neighbours(self, lat, lon, num, include_match=True):   # grid.cpp:72-75
        idx = self._closest(lat, lon, num, include_match)
        return np.stack([idx // self._nx, idx % self._nx], axis=1).astype(np.int32) if idx.size else np.zeros((0, 2), np.int32)

    def get_num_neighbours(self, lat, lon, radius, include_match=True):
        return len(self._neighbours(lat, lon, radius, include_match))

    def get_neighbours_with_distance(self, lat, lon, radius, include_match=True):   # grid.cpp:62-66
        idx, dist = self._neighbours(lat, lon, radius, include_match, True)
        ij = np.stack([idx // self._nx, idx % self._nx], axis=1).astype(np.int32) if idx.size else np.zeros((0, 2), np.int32)
        return ij, dist

    def get_point(self, y_index, x_index):   # grid.cpp:230-233
        i = int(y_index) * self._nx + int(x_index)
        f = [self._field(k)[i] for k in range(7)]
        return Point(f[0], f[1], f[2], f[3], self._type, f[4], f[5], f[6])

    def get_box(self, lat, lon):   # grid.cpp:149-229 -> [inside, Y1, X1, Y2, X2] (swig/gridpp.i:73-76 OUTPUT ints)
        qlat, qlon = np.array([lat], np.float32), np.array([lon], np.float32)
        inside, box = np.zeros(1, np.int32), np.zeros(4, np.int32)
        check(lib().gpp_grid_get_box(self._h, _ptr(qlat), _ptr(qlon), 1, _ptr(inside), _ptr(box)))
        return [bool(inside[0])] + [int(b) for b in box]

    def to_points(self):   # grid.cpp:131-145
        return Points(self._field(0), self._field(1), self._field(2), self._field(3), self._type)


# SWIG's flat names of the static KDTree methods (the reference's tests use them: tests/test_kdtree.py:56-58,111-112)
KDTree_calc_distance = KDTree.calc_distance
KDTree_calc_distance_fast = KDTree.calc_distance_fast
KDTree_calc_straight_distance = KDTree.calc_straight_distance
KDTree_deg2rad = KDTree.deg2rad
KDTree_rad2deg = KDTree.rad2deg


# ---- structure functions (include/gridpp.h:2069-2343, src/api/structure.cpp) -----------------------
_SK = dict(Barnes=0, Cressman=1, Soar=2, Toar=3, Powerlaw=4, Linear=5)
_ST_HAS_LOC, _ST_CV = 1, 2


class StructureFunction:
    """Base of the scalar structure functions; owns the gpp_structure descriptor handed to the C-ABI."""
    _s = None

    def _copy_struct(self):
        t = _capi.gpp_structure()
        C.memmove(C.byref(t), C.byref(self._s), C.sizeof(t))
        return t

    def localization_distance(self, p=None):
        d = C.c_float(0)
        lat, lon = (p.lat, p.lon) if p is not None else (0.0, 0.0)
        if p is None and self._s.field:
            raise ValueError("a spatially varying structure needs the point")
        check(lib().gpp_structure_localization_distance(C.byref(self._s), float(lat), float(lon), C.byref(d)))
        return d.value

    def _corr(self, p1, p2, background):
        if isinstance(p2, (list, tuple)):
            return np.array([self._corr(p1, q, background) for q in p2], np.float32)
        r = C.c_float(0)
        check(lib().gpp_structure_corr(C.byref(self._s), p1._seven(), p2._seven(), int(background), C.byref(r)))
        return r.value

    def corr(self, p1, p2):
        return self._corr(p1, p2, 0)

    def corr_background(self, p1, p2):
        return self._corr(p1, p2, 1)

    def clone(self):
        c = StructureFunction.__new__(type(self))
        c._s = self._copy_struct()
        c._field_owner = getattr(self, "_field_owner", None)   # keeps the HBM fields (and their grid) alive
        return c


class _Field:
    """Owns a gpp_field (h, v, w fields of a spatially varying structure resident in HBM)."""

    def __init__(self, grid, h, v, w, kind, min_rho):
        self.grid = grid   # the field borrows the grid handle
        self.h = C.c_void_p()
        check(lib().gpp_field_create(grid._h, _ptr(h), _ptr(v), _ptr(w), kind, float(min_rho), C.byref(self.h)))

    def __del__(self):
        try:
            if self.h:
                lib().gpp_field_destroy(self.h)
                self.h = None
        except Exception:
            pass


def _scalar_structure(obj, kind, h, v, w, hmax):
    if isinstance(h, Grid):   # (grid, h, v, w, min_rho): spatially varying form, e.g. src/api/structure.cpp:168-184
        grid, hf, vf, wf = h, v, w, hmax
        raise RuntimeError("internal: use _spatial_structure")
    for name, val in (("v", v), ("w", w)):   # e.g. structure.cpp:147-152
        if not is_valid(val) or val < 0:
            raise ValueError("%s must be >= 0" % name)
    mr = C.c_float(0)
    check(lib().gpp_structure_min_rho(kind, float(h), float(hmax), C.byref(mr)))
    obj._s = _capi.gpp_structure(kind, float(h), float(v), float(w), mr.value, 0, 0, 0.0, 0.0, 0, None)


def _spatial_structure(obj, kind, grid, h, v, w, min_rho):
    h, v, w = _vec(h, 2, "h"), _vec(v, 2, "v"), _vec(w, 2, "w")
    if h.shape == (1, 1) and v.shape == (1, 1) and w.shape == (1, 1):   # not spatial (structure.cpp:174-176)
        obj._s = _capi.gpp_structure(kind, float(h[0, 0]), float(v[0, 0]), float(w[0, 0]), float(min_rho), 0, 0, 0.0, 0.0, 0, None)
        return
    shape = tuple(grid.size())
    if h.shape != shape or v.shape != shape or w.shape != shape:
        raise ValueError("Grid size not the same as scale size")
    obj._field_owner = _Field(grid, h, v, w, kind, min_rho)
    obj._s = _capi.gpp_structure(kind, 0.0, 0.0, 0.0, float(min_rho), 0, 0, 0.0, 0.0, 0, obj._field_owner.h)


def _make_structure(obj, kind, args, has_hmax=True):
    """(h, v=0, w=0, hmax=MV) or (grid, h, v, w, min_rho=0.0013)"""
    if len(args) and isinstance(args[0], Grid):
        if len(args) < 4:
            raise TypeError("spatially varying structure: (grid, h, v, w, min_rho=0.0013)")
        _spatial_structure(obj, kind, args[0], args[1], args[2], args[3], args[4] if len(args) > 4 else 0.0013)
    else:
        a = list(args) + [0, 0, MV][len(args) - 1:] if len(args) < 4 else list(args)
        _scalar_structure(obj, kind, a[0], a[1], a[2], a[3] if has_hmax else MV)


class BarnesStructure(StructureFunction):
    """BarnesStructure(h, v=0, w=0, hmax=MV) (src/api/structure.cpp:143-167)"""

    def __init__(self, *args):
        _make_structure(self, _SK["Barnes"], args)


class CressmanStructure(StructureFunction):
    """CressmanStructure(h, v=0, w=0) (src/api/structure.cpp:287-312)"""

    def __init__(self, h, v=0, w=0):
        if not is_valid(h) or h < 0:   # StructureFunction(localization_distance) base ctor, structure.cpp:7-12
            raise ValueError("Structure function initizlied with invalid localization distance")
        _scalar_structure(self, _SK["Cressman"], h, v, w, MV)


class SoarStructure(StructureFunction):
    def __init__(self, *args):   # structure.cpp:317-341
        _make_structure(self, _SK["Soar"], args)


class ToarStructure(StructureFunction):
    def __init__(self, *args):   # structure.cpp:467-491
        _make_structure(self, _SK["Toar"], args)


class PowerlawStructure(StructureFunction):
    def __init__(self, *args):   # structure.cpp:618-642
        _make_structure(self, _SK["Powerlaw"], args)


class LinearStructure(StructureFunction):
    def __init__(self, *args):   # structure.cpp:765-789
        _make_structure(self, _SK["Linear"], args)


class MultipleStructure(StructureFunction):
    """MultipleStructure(structure_h, structure_v, structure_w) (src/api/structure.cpp:90-138): the horizontal factor
    (and the localization distance) comes from structure_h, the vertical one from structure_v, the land-area-fraction
    one from structure_w."""

    def __init__(self, structure_h, structure_v, structure_w):
        sh, sv, sw = structure_h._s, structure_v._s, structure_w._s
        kv = sv.kind_v - 1 if sv.kind_v else sv.kind
        kw = sw.kind_w - 1 if sw.kind_w else sw.kind
        # A component may itself be a MultipleStructure (the reference simply delegates, structure.cpp:90-138): structure_h then
        # contributes its horizontal part, structure_v the vertical part of ITS vertical component (kind_v, v, field_v) and
        # structure_w the laf part of its laf component.
        # spatially varying parts: the horizontal scale (and the localization distance) from structure_h's field, the vertical
        # scale from structure_v's, the laf scale from structure_w's (each at the nearest point of ITS grid to the first point)
        self._field_owner = [getattr(t, "_field_owner", None) for t in (structure_h, structure_v, structure_w)]
        if sh.field:
            loc, flags = 0.0, 0
        else:
            loc, flags = structure_h.localization_distance(), _ST_HAS_LOC
        fv = sv.field_v if sv.kind_v else sv.field
        fw = sw.field_w if sw.kind_w else sw.field
        self._s = _capi.gpp_structure(sh.kind, sh.h, sv.v, sw.w, sh.min_rho, kv + 1, kw + 1, loc, 0.0, flags, sh.field, fv, fw)


class CrossValidation(StructureFunction):
    """CrossValidation(structure, dist) (src/api/structure.cpp:910-944)"""

    def __init__(self, structure, dist):
        if not is_valid(dist) or dist < 0:
            raise ValueError("Invalid 'dist' in CrossValidation structure")
        self._s = structure._copy_struct()
        self._s.flags |= _ST_CV
        self._s.cv_dist = float(dist)
        self._field_owner = getattr(structure, "_field_owner", None)


def _structure(s):
    if not isinstance(s, StructureFunction) or s._s is None:
        raise RuntimeError("structure must be one of the gridpp_amd structure functions")
    return C.byref(s._s)


# ---- optimal interpolation (include/gridpp.h:162-248, src/api/oi.cpp) -----------------------------
def _oi_common(bg, background, bvariance, points, pobs, obs_variance, pbackground, bvariance_at_points,
               structure, max_points, allow_extrapolation, want_variance, deferred=False):
    if max_points < 0:
        raise ValueError("max_points must be >= 0")
    if not isinstance(bg, (Grid, Points)) or not isinstance(points, Points):
        raise TypeError("bgrid must be a Grid or Points, points a Points")
    if bg.get_coordinate_type() != points.get_coordinate_type():
        raise ValueError("Both background and observations points must be of same coordinate type (lat/lon or x/y)")
    nd = 2 if isinstance(bg, Grid) else 1
    f64 = _wants_f64(background, bvariance)
    dt = np.float64 if f64 else np.float32
    background = _vec(background, nd, "background", dt)
    shape = tuple(bg.size()) if nd == 2 else (bg.size(),)
    if _shape(background) != shape:
        raise ValueError("input field %s is not the same size as the grid %s" % (_shape(background), shape))
    if bvariance is not None:
        bvariance = _vec(bvariance, nd, "bvariance", dt)
        if _shape(bvariance) != shape:
            raise ValueError("Input bvariance is not the same size as the grid")
    S = points.size()
    pobs, obs_variance, pbackground = _vec(pobs, 1, "obs", dt), _vec(obs_variance, 1, "variance", dt), _vec(pbackground, 1, "background_at_points", dt)
    for name, a in (("Observations", pobs), ("Ratios", obs_variance), ("Background", pbackground)):
        if _shape(a)[0] != S:
            raise ValueError("%s (%d) and points (%d) size mismatch" % (name, _shape(a)[0], S))
    if bvariance_at_points is not None:
        bvariance_at_points = _vec(bvariance_at_points, 1, "bvariance_at_points", dt)
        if _shape(bvariance_at_points)[0] != S:
            raise ValueError("Background variance and points size mismatch")
    mem = _mem(background, bvariance, pobs, obs_variance, pbackground, bvariance_at_points)
    if mem == _capi.MEM_DEVICE:
        import torch
        torch.cuda.current_stream().synchronize()   # producers of the inputs ran on torch's stream
    elif f64:
        mem |= _capi.HOST_F64
    out = _empty_like_field(shape, background)
    var = _empty_like_field(shape, background) if want_variance else None
    if deferred:
        if mem != _capi.MEM_DEVICE:
            raise ValueError("optimal_interpolation_async takes device-resident fields (torch tensors on the GPU)")
        mem |= _capi.ASYNC
    rc = lib().gpp_optimal_interpolation_full(bg._h, _ptr(background), _ptr(bvariance), points._h, _ptr(pobs),
                                              _ptr(obs_variance), _ptr(pbackground), _ptr(bvariance_at_points),
                                              _structure(structure), int(max_points), int(bool(allow_extrapolation)),
                                              _ptr(out), _ptr(var), mem)
    if deferred:   # (the inputs, the handles and the structure stay referenced until the wait)
        # The library queues EVERY GPP_ASYNC call, also one that failed at submission (as a completed call that reports its error again): the
        # mirror keeps its queue aligned with that one -- the entry of a failed submission is simply dropped when it reaches the front.
        pending = PendingAnalysis(out, var, (bg, background, bvariance, points, pobs, obs_variance, pbackground, bvariance_at_points, structure))
        check(rc)
        return pending
    check(rc)
    return out, var


class PendingAnalysis:
    """An optimal_interpolation call that was enqueued with GPP_ASYNC (include/gridpp_hip.h: gpp_wait): wait() returns the analysis (and the
    variance, if asked for) once it is complete and raises what the call would have raised.  The library completes its deferred calls in
    order: waiting for a later one first completes the earlier ones."""
    _tls = None      # per thread: the library queues deferred calls per calling thread (gpp_wait completes the oldest one of ITS thread)

    @classmethod
    def _q(cls):
        import threading
        if cls._tls is None:
            cls._tls = threading.local()
        if not hasattr(cls._tls, "q"):
            cls._tls.q = []
        return cls._tls.q

    def __init__(self, out, var, keep):
        self._out, self._var, self._keep, self._rc, self._msg = out, var, keep, None, None
        self._queue = PendingAnalysis._q()      # (wait() from another thread than the one that made the call is an error of the caller)
        self._queue.append(self)

    def _complete_front(self):
        front = self._queue.pop(0)
        front._rc = lib().gpp_wait()
        front._msg = lib().gpp_last_error().decode("utf-8", "replace") if front._rc != _capi.GPP_OK else None
        front._stats = oi_last_stats() if front._rc == _capi.GPP_OK else None
        front._keep = None

    def wait(self):
        while self._rc is None:
            self._complete_front()
        if self._rc == _capi.GPP_EINVAL:
            raise ValueError(self._msg)
        if self._rc != _capi.GPP_OK:
            raise RuntimeError(self._msg)
        return self._out if self._var is None else (self._out, self._var)

    def stats(self):
        """gpp_oi_last_stats of THIS call (valid after wait())"""
        self.wait()
        return self._stats


def optimal_interpolation_async(bgrid, background, points, pobs, pratios, pbackground, structure, max_points, allow_extrapolation=True):
    """optimal_interpolation on device-resident fields without waiting for the result: returns a PendingAnalysis.  For a caller that streams
    analyses through one GPU (one per observation set; bench.py, gridpp_amd.dist): in the steady state of such a stream -- the same Grid and
    Points objects as the call before -- the library needs the host for nothing and the next call can be enqueued while this one runs.
    The tensors passed in must not be modified before wait() returns."""
    return _oi_common(bgrid, background, None, points, pobs, pratios, pbackground, None, structure, max_points, allow_extrapolation, False, deferred=True)


def optimal_interpolation(bgrid, background, points, pobs, pratios, pbackground, structure, max_points, allow_extrapolation=True):
    """gridpp::optimal_interpolation, Grid (src/api/oi.cpp:26-87) and Points (:89-136) overloads."""
    return _oi_common(bgrid, background, None, points, pobs, pratios, pbackground, None, structure, max_points,
                      allow_extrapolation, False)[0]


def optimal_interpolation_full_async(bgrid, background, bvariance, points, obs, obs_variance, background_at_points,
                                     bvariance_at_points, structure, max_points, allow_extrapolation=True):
    """optimal_interpolation_full without waiting for the result (see optimal_interpolation_async): PendingAnalysis.wait() returns
    (analysis, analysis_variance)."""
    return _oi_common(bgrid, background, bvariance, points, obs, obs_variance, background_at_points, bvariance_at_points, structure, max_points,
                      allow_extrapolation, True, deferred=True)


def optimal_interpolation_full(bgrid, background, bvariance, points, obs, obs_variance, background_at_points,
                               bvariance_at_points, structure, max_points, allow_extrapolation=True):
    """gridpp::optimal_interpolation_full (src/api/oi.cpp:138-412); returns (analysis, analysis_variance)."""
    out, var = _oi_common(bgrid, background, bvariance, points, obs, obs_variance, background_at_points,
                          bvariance_at_points, structure, max_points, allow_extrapolation, True)
    return out, var


def oi_last_stats():
    s = _capi.gpp_oi_stats()
    check(lib().gpp_oi_last_stats(C.byref(s)))
    return dict(cells=s.cells, cells_updated=s.cells_updated, solves=s.solves, fallback_tiles=s.fallback_tiles, kernel_ms=s.kernel_ms,
                union_kernel_ms=s.union_kernel_ms, fallback_subtiles=s.fallback_subtiles, big_cells=s.big_cells)


# ---- nearest (src/api/nearest.cpp:124-144) ----------------------------------------------------------
def nearest(igrid, opoints, values):
    """All eight overloads of src/api/nearest.cpp: Grid|Points -> Grid|Points, with or without a leading time dimension."""
    nd = 2 if isinstance(igrid, Grid) else 1
    if _is_dev(values):
        import torch
        values = values.contiguous().to(torch.float32)
    else:
        values = np.ascontiguousarray(np.asarray(values), dtype=np.float64 if _wants_f64(values) else np.float32)
        if values.size == 0 and values.ndim < nd:
            values = values.reshape((0,) * nd)
    shp = _shape(values)
    if len(shp) not in (nd, nd + 1):
        raise RuntimeError("values must have %d or %d dimensions" % (nd, nd + 1))
    levels = len(shp) == nd + 1
    ishape = tuple(igrid.size()) if nd == 2 else (igrid.size(),)
    if nd == 2:    # src/api/util.cpp:427-432
        empty = shp[0] == 0 or (levels and shp[1] == 0)
    else:          # util.cpp:433-438: a vec2 with no time levels passes, a vec must match
        empty = levels and shp[0] == 0
    if not empty and tuple(shp[-nd:]) != ishape:
        raise ValueError("Grid size is not the same as values" if nd == 2 else "Points size is not the same as values")
    oshape = tuple(opoints.size()) if isinstance(opoints, Grid) else (opoints.size(),)
    nt = shp[0] if levels else 1
    lead = (nt,) if levels else ()
    out = _empty_like_field(lead + oshape, values)
    if int(np.prod(lead + oshape)) == 0:
        return out
    if igrid._n and empty:
        raise ValueError("Grid size is not the same as values")
    mem = _mem(values)
    _sync_if_dev(mem)
    if not _is_dev(values) and values.dtype == np.float64:
        mem |= _capi.HOST_F64
    check(lib().gpp_nearest_levels(igrid._h, opoints._h, _ptr(values), nt, _ptr(out), mem))
    return out


# ---- count / gridding (include/gridpp.h:938-1010, src/api/count.cpp, src/api/gridding.cpp) -------------
def _out_shape(o):
    return tuple(o.size()) if isinstance(o, Grid) else (o.size(),)


def count(ipoints, opoints, radius):
    """Number of points of `ipoints` (Grid or Points) within `radius` of every location of `opoints` (count.cpp:6-66)."""
    out = np.empty(_out_shape(opoints), np.float32)
    if out.size:
        check(lib().gpp_count(ipoints._h, opoints._h, float(radius), _ptr(out), _capi.MEM_HOST))
    return out


def distance(ipoints, opoints, num):
    """Largest distance from every location of `opoints` to its `num` nearest points of `ipoints` (distance.cpp:6-120)."""
    if ipoints.get_coordinate_type() != opoints.get_coordinate_type():
        raise ValueError("Incompatible coordinate types")
    out = np.empty(_out_shape(opoints), np.float32)
    if out.size:
        check(lib().gpp_distance(ipoints._h, opoints._h, int(num), 0 if isinstance(opoints, Grid) else 1, _ptr(out), _capi.MEM_HOST))
    return out


def staticcorr_points(points, knots, structure, max_points):
    """(L, K) static correlations between a set of points and a set of knots (src/api/corr_points.cpp:26-131)."""
    if max_points < 0:
        raise ValueError("max_points must be >= 0")
    if points.get_coordinate_type() != knots.get_coordinate_type():
        raise ValueError("Both background grid and observations points must be of same coordinate type (lat/lon or x/y)")
    out = _host_empty((points.size(), knots.size()))
    if out.size:
        check(lib().gpp_staticcorr_points(points._h, knots._h, _structure(structure), int(max_points), _ptr(out), _capi.MEM_HOST))
    return out


def _point_values(ipoints, values):
    if _is_dev(values):
        import torch
        values = values.contiguous().to(torch.float32)
        if values.dim() != 1:
            raise RuntimeError("values must have 1 dimension")
    else:
        values = _vec(values, 1, "values")
    if _shape(values)[0] != ipoints.size():
        raise ValueError("Points size is not the same as values")
    return values


def gridding(grid, points, values, radius, min_num, statistic):
    """gridding.cpp:6-63; `grid` may be a Grid or a Points (output shaped accordingly)."""
    values = _point_values(points, values)
    if not is_valid(radius) or radius < 0:
        raise ValueError("radius must be >= 0")
    if min_num < 0:
        raise ValueError("min_num must be >= 0")
    out = _empty_like_field(_out_shape(grid), values)
    if int(np.prod(_out_shape(grid))):
        mem = _mem(values)
        _sync_if_dev(mem)
        check(lib().gpp_gridding(grid._h, points._h, _ptr(values), float(radius), int(min_num), int(statistic), _ptr(out), mem))
    return out


def gridding_nearest(grid, points, values, min_num, statistic):
    """gridding.cpp:65-131"""
    values = _point_values(points, values)
    if min_num < 0:
        raise ValueError("min_num must be >= 0")
    out = _empty_like_field(_out_shape(grid), values)
    if int(np.prod(_out_shape(grid))) or points.size():
        mem = _mem(values)
        _sync_if_dev(mem)
        check(lib().gpp_gridding_nearest(grid._h, points._h, _ptr(values), int(min_num), int(statistic), _ptr(out), mem))
    return out


# ---- fill / doping / neighbourhood_search / calc_gradient -----------------------------------------------
MinMax, LinearRegression = 0, 10   # include/gridpp.h:126-129


def _grid_field(igrid, values, what="values"):
    values = _vec(values, 2, what)
    shp = _shape(values)
    if shp[0] != 0 and tuple(shp) != tuple(igrid.size()):       # src/api/util.cpp:427-429
        raise ValueError("Grid size is not the same as " + what)
    return values


def fill(igrid, input, points, radii, value, outside):
    """src/api/fill.cpp:6-41"""
    input = _grid_field(igrid, input)
    radii = np.ascontiguousarray(np.asarray(radii, np.float32).ravel())
    if radii.size != points.size():
        raise ValueError("Points size is not the same as radii size")
    out = _empty_like_field(_shape(input), input)
    if int(np.prod(_shape(input))):
        mem = _mem(input)
        _sync_if_dev(mem)
        check(lib().gpp_fill(igrid._h, _ptr(input), points._h, _ptr(radii), float(value), int(bool(outside)), _ptr(out), mem))
    return out


def fill_missing(values):
    """src/api/fill.cpp:43-134"""
    f64 = _wants_f64(values)
    values = _vec(values, 2, "values", np.float64 if f64 else np.float32)
    out = _empty_like_field(_shape(values), values)
    ny, nx = _shape(values)
    if ny * nx:
        mem = _mem(values)
        _sync_if_dev(mem)
        check(lib().gpp_fill_missing(_ptr(values), ny, nx, _ptr(out), mem | (_capi.HOST_F64 if f64 else 0)))
    return out


def _doping(igrid, background, points, observations, halfwidth, radii, max_elev_diff):
    background = _grid_field(igrid, background, "observations")
    observations = _vec(observations, 1, "observations")
    if _shape(observations)[0] != points.size():
        raise ValueError("Points size is not the same as observations size")
    per = halfwidth if halfwidth is not None else radii
    if per.size != points.size():
        raise ValueError("Points size is not the same as %s size" % ("halfwidth" if halfwidth is not None else "radii"))
    out = _empty_like_field(_shape(background), background)
    if int(np.prod(_shape(background))):
        mem = _mem(background, observations)
        _sync_if_dev(mem)
        check(lib().gpp_doping(igrid._h, _ptr(background), points._h, _ptr(observations), _ptr(halfwidth), _ptr(radii), float(max_elev_diff),
                               _ptr(out), mem))
    return out


def doping_square(igrid, background, points, observations, halfwidth, max_elev_diff=MV):
    """src/api/doping.cpp:5-48"""
    return _doping(igrid, background, points, observations, np.ascontiguousarray(np.asarray(halfwidth, np.int32).ravel()), None, max_elev_diff)


def doping_circle(igrid, background, points, observations, radii, max_elev_diff=MV):
    """src/api/doping.cpp:50-93"""
    return _doping(igrid, background, points, observations, None, np.ascontiguousarray(np.asarray(radii, np.float32).ravel()), max_elev_diff)


def neighbourhood_search(array, search_array, halfwidth, search_target_min, search_target_max, search_delta, apply_array=None):
    """src/api/neighbourhood_search.cpp:7-113"""
    f64 = _wants_f64(array, search_array)
    dt = np.float64 if f64 else np.float32
    array, search_array = _vec(array, 2, "array", dt), _vec(search_array, 2, "search_array", dt)
    if _shape(array) != _shape(search_array):
        raise ValueError("search_array must either be the same size as array")
    if search_target_min > search_target_max:
        raise ValueError("Search_target_min must be smaller than search_target_max")
    if halfwidth < 0:
        raise ValueError("halfwidth must be positive")
    ap = None
    if apply_array is not None and np.size(apply_array) > 0:
        if _is_dev(apply_array):
            import torch
            ap = apply_array.contiguous().to(torch.int32)
        else:
            ap = np.ascontiguousarray(np.asarray(apply_array).astype(np.int32))
        # (the reference indexes apply_array[y][x] for every cell whatever its size; anything but a full-size mask reads out of bounds there)
        if tuple(ap.shape) != _shape(array):
            raise ValueError("apply_array must either be empty or same size as array")
    out = _empty_like_field(_shape(array), array)
    ny, nx = _shape(array)
    if ny * nx:
        mem = _mem(array, search_array, ap)
        _sync_if_dev(mem)
        check(lib().gpp_neighbourhood_search(_ptr(array), _ptr(search_array), ny, nx, int(halfwidth), float(search_target_min),
                                             float(search_target_max), float(search_delta), _ptr(ap), _ptr(out),
                                             mem | (_capi.HOST_F64 if f64 and mem == _capi.MEM_HOST else 0)))
    return out


def calc_gradient(base, values, gradient_type, halfwidth, num_min=2, min_range=MV, default_gradient=0):
    """src/api/calc_gradient.cpp:7-126"""
    f64 = _wants_f64(base, values)
    dt = np.float64 if f64 else np.float32
    base, values = _vec(base, 2, "base", dt), _vec(values, 2, "values", dt)
    if halfwidth <= 0:
        raise ValueError("Halwidth cannot be <= 0; must be positive integer")
    if is_valid(min_range) and min_range < 0:
        raise ValueError("min_range must be >= 0")
    if num_min < 0:
        raise ValueError("num_min must be >= 0")
    if _shape(base)[0] == 0:
        raise ValueError("base input has no size")
    if _shape(base) != _shape(values):
        raise ValueError("base is not the same size as values")
    out = _empty_like_field(_shape(base), base)
    ny, nx = _shape(base)
    mem = _mem(base, values)
    _sync_if_dev(mem)
    check(lib().gpp_calc_gradient(_ptr(base), _ptr(values), ny, nx, int(gradient_type), int(halfwidth), int(num_min), float(min_range),
                                  float(default_gradient), _ptr(out), mem | (_capi.HOST_F64 if f64 and mem == _capi.MEM_HOST else 0)))
    return out


# ---- bilinear (include/gridpp.h:902-930, src/api/bilinear.cpp:26-135) ---------------------------------
def bilinear(igrid, opoints, values):
    """values (Y, X) -> output shaped like opoints; values (T, Y, X) -> (T,) + that shape."""
    if not isinstance(igrid, Grid):
        raise TypeError("bilinear: the input must be a Grid")
    if _is_dev(values):
        import torch
        values = values.contiguous().to(torch.float32)
    else:
        values = np.ascontiguousarray(np.asarray(values), dtype=np.float64 if _wants_f64(values) else np.float32)
    shp = _shape(values)
    if len(shp) not in (2, 3):
        raise RuntimeError("bilinear: values must be 2-D or 3-D")
    # src/api/util.cpp:427-432: no rows (2-D) / no time levels or no rows (3-D) passes the size check
    empty = shp[0] == 0 or (len(shp) == 3 and shp[1] == 0)
    if not empty and tuple(shp[-2:]) != tuple(igrid.size()):
        raise ValueError("Grid size is not the same as values")
    oshape = tuple(opoints.size()) if isinstance(opoints, Grid) else (opoints.size(),)
    nt = shp[0] if len(shp) == 3 else 1
    lead = (nt,) if len(shp) == 3 else ()
    out = _empty_like_field(lead + oshape, values)
    if int(np.prod(lead + oshape)) == 0:
        return out
    if igrid._n and empty:
        raise ValueError("Grid size is not the same as values")   # nothing to read from (the reference would index past the end)
    mem = _mem(values)
    _sync_if_dev(mem)
    if not _is_dev(values) and values.dtype == np.float64:
        mem |= _capi.HOST_F64
    check(lib().gpp_bilinear(igrid._h, opoints._h, _ptr(values), nt, _ptr(out), mem))
    return out


def point_in_rectangle(A, B, C_, D, m):   # src/api/util.cpp:571-582
    corners = (C.c_float * 8)(A.lat, A.lon, B.lat, B.lon, C_.lat, C_.lon, D.lat, D.lon)
    inside = C.c_int(0)
    check(lib().gpp_point_in_rectangle(corners, float(m.lat), float(m.lon), C.byref(inside)))
    return bool(inside.value)


# ---- neighbourhood filters (include/gridpp.h:588-716, src/api/neighbourhood.cpp) ---------------------
def _field23(a, name="input", keep_f64=False):
    """2-D or 3-D field -> (array, ny, nx, ne, is3d); [[]] -> empty.  keep_f64: a large float64 array stays float64 (the caller
    passes GPP_HOST_F64)."""
    if _is_dev(a):
        nd = a.dim()
        arr = _vec(a, nd, name)
    else:
        arr = np.ascontiguousarray(np.asarray(a), dtype=np.float64 if keep_f64 and _wants_f64(a) else np.float32)
        nd = arr.ndim
    if nd not in (2, 3):
        if getattr(arr, "size", 1) == 0:
            return arr, 0, 0, 0, 0
        raise RuntimeError("%s must have 2 or 3 dimensions" % name)
    shp = tuple(arr.shape)
    ny, nx = shp[0], shp[1]
    ne = shp[2] if nd == 3 else 1
    return arr, ny, nx, ne, int(nd == 3)


def _sync_if_dev(mem):
    if mem == _capi.MEM_DEVICE:
        import torch
        torch.cuda.current_stream().synchronize()


def neighbourhood(input, halfwidth, statistic):
    """gridpp::neighbourhood for 2-D and 3-D (Y, X, E) input (src/api/neighbourhood.cpp:12-242)."""
    arr, ny, nx, ne, is3d = _field23(input, keep_f64=True)
    if halfwidth < 0:
        raise ValueError("Half width must be > 0")
    if statistic == Quantile:
        raise ValueError("Use neighbourhood_quantile for computing neighbourhood quantiles")
    if ny * nx * ne == 0:
        return np.zeros((0, 0), np.float32)
    mem = _mem(arr)
    _sync_if_dev(mem)
    if not _is_dev(arr) and arr.dtype == np.float64:
        mem |= _capi.HOST_F64
    out = _empty_like_field((ny, nx), arr)
    check(lib().gpp_neighbourhood(_ptr(arr), ny, nx, ne, is3d, int(halfwidth), int(statistic), _ptr(out), mem))
    return out


def neighbourhood_brute_force(input, halfwidth, statistic):
    arr, ny, nx, ne, is3d = _field23(input)
    if halfwidth < 0:
        raise ValueError("Half width must be > 0")
    if ny * nx * ne == 0:
        return np.zeros((0, 0), np.float32)
    mem = _mem(arr)
    _sync_if_dev(mem)
    out = _empty_like_field((ny, nx), arr)
    check(lib().gpp_neighbourhood_brute_force(_ptr(arr), ny, nx, ne, int(halfwidth), int(statistic), 0.0, _ptr(out), mem))
    return out


def neighbourhood_quantile(input, quantile, halfwidth):
    """Exact neighbourhood quantile (src/api/neighbourhood.cpp:534-539)."""
    arr, ny, nx, ne, is3d = _field23(input)
    if halfwidth < 0:
        raise ValueError("Half width must be > 0")
    if ny * nx * ne == 0:
        return np.zeros((0, 0), np.float32)
    mem = _mem(arr)
    _sync_if_dev(mem)
    out = _empty_like_field((ny, nx), arr)
    check(lib().gpp_neighbourhood_brute_force(_ptr(arr), ny, nx, ne, int(halfwidth), Quantile, float(quantile), _ptr(out), mem))
    return out


def neighbourhood_quantile_fast(input, quantile, halfwidth, thresholds):
    """All four overloads of gridpp::neighbourhood_quantile_fast (src/api/neighbourhood.cpp:296-527)."""
    arr, ny, nx, ne, is3d = _field23(input, keep_f64=True)
    if halfwidth < 0:
        raise ValueError("Half width must be > 0")
    if ny * nx * ne == 0:
        return np.zeros((0, 0), np.float32)
    mem = _mem(arr)
    if np.ndim(quantile) == 0 and not _is_dev(quantile):
        q = np.array([quantile], np.float32)
    else:
        q = _vec(quantile, 2, "quantile")
        if _shape(q) not in ((1, 1), (ny, nx)):
            raise ValueError("Quantile must be the same size as input, or size (1, 1)")
    thr = _vec(thresholds, 1, "thresholds")
    nq = int(np.prod(_shape(q)))
    q_host = False
    if mem == _capi.MEM_DEVICE:
        import torch
        if not _is_dev(q) and nq == 1:
            q_host = True        # (a scalar beside a device-resident field stays on the host: GPP_Q_HOST)
            q = np.ascontiguousarray(q, np.float32)
        else:
            q = q if _is_dev(q) else torch.from_numpy(np.asarray(q)).to(arr.device)
        thr = thr if _is_dev(thr) else torch.from_numpy(np.asarray(thr)).to(arr.device)
    _sync_if_dev(mem)
    out = _empty_like_field((ny, nx), arr)
    if not _is_dev(arr) and arr.dtype == np.float64:
        mem |= _capi.HOST_F64
    if q_host:
        mem |= _capi.Q_HOST
    check(lib().gpp_neighbourhood_quantile_fast(_ptr(arr), ny, nx, ne, is3d, _ptr(q), nq, int(halfwidth), _ptr(thr),
                                                int(_shape(thr)[0]), _ptr(out), mem))
    return out


def neighbourhood_ens(input, halfwidth, statistic):   # deprecated aliases (neighbourhood.cpp:541-552), with the reference's message
    future_deprecation_warning("neighbourhood_ens", "neighbourhood")
    return neighbourhood(input, halfwidth, statistic)


def neighbourhood_quantile_ens(input, quantile, halfwidth):
    future_deprecation_warning("neighbourhood_quantile_ens", "neighbourhood_quantile")
    return neighbourhood_quantile(input, quantile, halfwidth)


def neighbourhood_quantile_ens_fast(input, quantile, halfwidth, thresholds):
    future_deprecation_warning("neighbourhood_quantile_ens_fast", "neighbourhood_quantile_fast")
    return neighbourhood_quantile_fast(input, quantile, halfwidth, thresholds)


# ---- util (include/gridpp.h:1454-1482, src/api/util.cpp) ---------------------------------------------
def calc_statistic(array, statistic):
    """gridpp::calc_statistic for a vector (-> float) or a 2-D array (-> one value per row)."""
    a = np.ascontiguousarray(np.asarray(array), dtype=np.float32)
    if a.ndim not in (1, 2):
        raise RuntimeError("array must have 1 or 2 dimensions")
    rows = 1 if a.ndim == 1 else a.shape[0]
    length = a.shape[-1]
    out = np.empty(rows, np.float32)
    if rows:
        check(lib().gpp_calc_statistic(_ptr(a), rows, length, int(statistic), _ptr(out), _capi.MEM_HOST))
    return float(out[0]) if a.ndim == 1 else out


def calc_quantile(array, quantile):
    """gridpp::calc_quantile: (vec, q) -> float, (vec2, q) -> vec, (vec3, vec2 q) -> vec2."""
    a = np.ascontiguousarray(np.asarray(array), dtype=np.float32)
    q = np.ascontiguousarray(np.asarray(quantile), dtype=np.float32)
    if a.ndim == 3:
        if q.shape != a.shape[:2]:
            raise ValueError("Dimension mismatch between array and quantile")
        if a.shape[0] * a.shape[1] == 0:
            return np.zeros((0, 0), np.float32)
        rows, length = a.shape[0] * a.shape[1], a.shape[2]
        out = np.empty(rows, np.float32)
        check(lib().gpp_calc_quantile(_ptr(a), rows, length, _ptr(q), rows, _ptr(out), _capi.MEM_HOST))
        return out.reshape(a.shape[:2])
    if a.ndim not in (1, 2):
        raise RuntimeError("array must have 1, 2 or 3 dimensions")
    rows = 1 if a.ndim == 1 else a.shape[0]
    out = np.empty(rows, np.float32)
    qq = q.reshape(1)
    if rows:
        check(lib().gpp_calc_quantile(_ptr(a), rows, a.shape[-1], _ptr(qq), 1, _ptr(out), _capi.MEM_HOST))
    return float(out[0]) if a.ndim == 1 else out


def _even_quantiles(values, num, only_valid):
    dev = _is_dev(values)
    v = values.contiguous().float().reshape(-1) if dev else np.ascontiguousarray(np.asarray(values), dtype=np.float32).ravel()
    n = int(v.numel()) if dev else v.size
    out = np.empty(max(int(num), 1) if n else 1, np.float32)
    if num > 0 and n > 0 and num >= n:
        out = np.empty(n, np.float32)
    cnt = C.c_int(0)
    if dev:
        import torch
        torch.cuda.current_stream().synchronize()
        dout = torch.empty(out.size, dtype=torch.float32, device=v.device)
        check(lib().gpp_calc_even_quantiles(_ptr(v), n, int(num), int(only_valid), _ptr(dout), C.byref(cnt), _capi.MEM_DEVICE))
        return dout[:cnt.value].cpu().numpy()
    check(lib().gpp_calc_even_quantiles(_ptr(v), n, int(num), int(only_valid), _ptr(out), C.byref(cnt), _capi.MEM_HOST))
    return out[:cnt.value].copy()


def calc_even_quantiles(values, num):
    """gridpp::calc_even_quantiles (src/api/util.cpp:261-338)."""
    return _even_quantiles(values, num, 0)


def get_neighbourhood_thresholds(input, num_thresholds):
    """gridpp::get_neighbourhood_thresholds for 2-D / 3-D input (src/api/neighbourhood.cpp:243-295)."""
    if num_thresholds <= 0:
        raise ValueError("num_thresholds must be > 0")
    return _even_quantiles(input, num_thresholds, 1)


# ---- ensemble OI (include/gridpp.h:263-294, src/api/oi_ensi.cpp) ----------------------------------------
def optimal_interpolation_ensi(bgrid, background, points, pobs, psigmas, pbackground, structure, max_points, allow_extrapolation=True):
    """gridpp::optimal_interpolation_ensi, Grid (background (Y, X, E)) and Points (background (N, E)) overloads."""
    if max_points < 0:
        raise ValueError("max_points must be >= 0")
    if not isinstance(bgrid, (Grid, Points)) or not isinstance(points, Points):
        raise TypeError("bgrid must be a Grid or Points, points a Points")
    nd = 3 if isinstance(bgrid, Grid) else 2
    S = points.size()
    f64 = S > 0 and _wants_f64(background)
    dt = np.float64 if f64 else np.float32
    background = _vec(background, nd, "background", dt)
    if S == 0:   # src/api/oi_ensi.cpp:48-50,135-137: returns the background before any other check
        return background.clone() if _is_dev(background) else background.copy()
    if bgrid.get_coordinate_type() != points.get_coordinate_type():
        raise ValueError("Both background and observations points must be of same coordinate type (lat/lon or x/y)")
    shape = tuple(bgrid.size()) if nd == 3 else (bgrid.size(),)
    if nd == 3 and shape[0] * shape[1] == 0:
        raise ValueError("Grid size cannot be zero")
    if _shape(background)[:nd - 1] != shape:
        raise ValueError("Input field is not the same size as the grid")
    E = _shape(background)[-1]
    pobs, psigmas = _vec(pobs, 1, "obs", dt), _vec(psigmas, 1, "sigmas", dt)
    pbackground = _vec(pbackground, 2, "background_at_points", dt)
    if _shape(pobs)[0] != S:
        raise ValueError("Observations and points size mismatch")
    if _shape(psigmas)[0] != S:
        raise ValueError("Sigmas and points size mismatch")
    if _shape(pbackground)[0] != S:
        raise ValueError("Background and points size mismatch")
    if _shape(pbackground)[1] != E:
        raise ValueError("Ensemble members in gridded background is not the same as in the point background")
    mem = _mem(background, pobs, psigmas, pbackground)
    _sync_if_dev(mem)
    if f64 and mem == _capi.MEM_HOST:
        mem |= _capi.HOST_F64
    out = _empty_like_field(_shape(background), background)
    check(lib().gpp_optimal_interpolation_ensi(bgrid._h, _ptr(background), int(E), points._h, _ptr(pobs), _ptr(psigmas),
                                               _ptr(pbackground), _structure(structure), int(max_points),
                                               int(bool(allow_extrapolation)), _ptr(out), mem))
    _ensi_warnings()
    return out


def ensi_last_stats():
    """cells, the grid points the last optimal_interpolation_ensi left at their background values (singular / non-finite E x E system:
    the reference's num_condition_warning, oi_ensi.cpp:153,386-390), its second counter (always 0 here) and the kernel time"""
    s = _capi.gpp_ensi_stats()
    check(lib().gpp_ensi_last_stats(C.byref(s)))
    return dict(cells=s.cells, condition_passthrough=s.condition_passthrough, real_part_passthrough=s.real_part_passthrough, kernel_ms=s.kernel_ms)


def _ensi_warnings():
    """the two warnings the reference prints at the end of optimal_interpolation_ensi (oi_ensi.cpp:557-566)"""
    s = ensi_last_stats()
    if s["condition_passthrough"] > 0:
        warning("Condition number error in %d points. Using raw values in those points." % s["condition_passthrough"])
    if s["real_part_passthrough"] > 0:
        warning("Could not find the real part of W in %d points. Using raw values in those points." % s["real_part_passthrough"])


def _ensi_multi(variant, name, bgrid, bratios, background, background_corr, points, pobs, pratios, pbackground, pbackground_corr,
                structure, max_points, allow_extrapolation):
    """Shared body of the optimal_interpolation_ensi_multi_* mirrors (validation as src/api/oi_ensi_multi.cpp:341-362)."""
    if max_points < 0:
        raise ValueError("max_points must be >= 0")
    if not isinstance(bgrid, (Grid, Points)) or not isinstance(points, Points):
        raise TypeError("bgrid must be a Grid or Points, points a Points")
    nd = 3 if isinstance(bgrid, Grid) else 2
    S = points.size()
    background = _vec(background, nd, "background", np.float32)
    if S == 0:
        return background.clone() if _is_dev(background) else background.copy()
    if bgrid.get_coordinate_type() != points.get_coordinate_type():
        raise ValueError("Both background and observations points must be of same coorindate type (lat/lon or x/y)")
    shape = tuple(bgrid.size()) if nd == 3 else (bgrid.size(),)
    if nd == 3 and shape[0] * shape[1] == 0:
        raise ValueError("Grid size cannot be zero")
    if _shape(background)[:nd - 1] != shape:
        raise ValueError("Input background field is not the same size as the grid")
    E = _shape(background)[-1]
    corr = variant != 2
    if corr:
        background_corr = _vec(background_corr, nd, "background_corr", np.float32)
        if _shape(background_corr) != _shape(background):
            raise ValueError("Input background_corr field is not the same size as the grid")
        pbackground_corr = _vec(pbackground_corr, 2, "pbackground_corr", np.float32)
        if _shape(pbackground_corr)[0] != S:
            raise ValueError("Background_corr and points size mismatch")
    bratios = _vec(bratios, nd - 1, "bratios", np.float32)
    if _shape(bratios) != shape:
        raise ValueError("Bratios and grid size mismatch")
    pobs = _vec(pobs, 1 if variant == 3 else 2, "pobs", np.float32)
    pratios = _vec(pratios, 1, "pratios", np.float32)
    pbackground = _vec(pbackground, 2, "pbackground", np.float32)
    if _shape(pobs)[0] != S:
        raise ValueError("Observations and points exception mismatch")
    if _shape(pratios)[0] != S:
        raise ValueError("Pratios and points size mismatch")
    if _shape(pbackground)[0] != S:
        raise ValueError("Background and points size mismatch")
    if _shape(pbackground)[1] != E or (variant != 3 and _shape(pobs)[1] != E) or (corr and _shape(pbackground_corr)[1] != E):
        raise ValueError("Ensemble members in gridded background is not the same as in the point fields")
    args = [bratios, background, pobs, pratios, pbackground] + ([background_corr, pbackground_corr] if corr else [])
    mem = _mem(*args)
    _sync_if_dev(mem)
    out = _empty_like_field(_shape(background), background)
    check(lib().gpp_optimal_interpolation_ensi_multi(int(variant), bgrid._h, _ptr(bratios), _ptr(background),
                                                     _ptr(background_corr) if corr else None, int(E), points._h, _ptr(pobs), _ptr(pratios),
                                                     _ptr(pbackground), _ptr(pbackground_corr) if corr else None, _structure(structure),
                                                     int(max_points), int(bool(allow_extrapolation)), _ptr(out), mem))
    return out


def optimal_interpolation_ensi_multi_ebe(bgrid, bratios, background, background_corr, obs_points, pobs, pratios, pbackground,
                                         pbackground_corr, structure, max_points, allow_extrapolation=True):
    """gridpp::optimal_interpolation_ensi_multi_ebe (include/gridpp.h:311-322,373-384): ensemble-based correlations, member by member."""
    return _ensi_multi(1, "ebe", bgrid, bratios, background, background_corr, obs_points, pobs, pratios, pbackground, pbackground_corr,
                       structure, max_points, allow_extrapolation)


def optimal_interpolation_ensi_multi_ebesc(bgrid, bratios, background, obs_points, pobs, pratios, pbackground, structure, max_points,
                                           allow_extrapolation=True):
    """gridpp::optimal_interpolation_ensi_multi_ebesc (include/gridpp.h:336-345,398-407): static correlations, member by member."""
    return _ensi_multi(2, "ebesc", bgrid, bratios, background, None, obs_points, pobs, pratios, pbackground, None, structure, max_points,
                       allow_extrapolation)


def optimal_interpolation_ensi_multi_utem(bgrid, bratios, background, background_corr, obs_points, pobs, pratios, pbackground,
                                          pbackground_corr, structure, max_points, allow_extrapolation=True):
    """gridpp::optimal_interpolation_ensi_multi_utem (include/gridpp.h:360-371,421-432): ensemble mean first, then the analysis ensemble."""
    return _ensi_multi(3, "utem", bgrid, bratios, background, background_corr, obs_points, pobs, pratios, pbackground, pbackground_corr,
                       structure, max_points, allow_extrapolation)


def ensi_last_kernel_ms():
    ms = C.c_float(0)
    check(lib().gpp_ensi_last_kernel_ms(C.byref(ms)))
    return ms.value


def ensi_set_convergence(to_convergence):
    """True: the per-cell Jacobi sweeps of optimal_interpolation_ensi run to convergence (about twice the time);
    False (default): they stop early and a perturbation series supplies the rest."""
    check(lib().gpp_ensi_set_convergence(1 if to_convergence else 0))


def set_path_override(name, value):
    """The library's one test / A-B hook: `name` = a GPP_* path switch (they select between implementations with identical results),
    `value` = its setting as a string, None clears it.  The library reads no environment variable itself."""
    check(lib().gpp_set_path_override(str(name).encode(), None if value is None else str(value).encode()))


def active_overrides():
    """Names of the GPP_* path switches that are set (they select implementations, never results)."""
    buf = C.create_string_buffer(4096)
    n = lib().gpp_active_overrides(buf, len(buf))
    return [s for s in buf.value.decode().split(",") if s] if n > 0 else []


def release_workspaces():
    """Frees the large device workspaces this thread keeps between calls."""
    check(lib().gpp_release_workspaces())


# ---- the typemap test helpers of the reference (src/api/swig.cpp:6-100, include/gridpp.h:1680-1702): they exist so that
# tests/test_swig.py can pin the binding layer -- element types of results, accepted input types, ndim checks, zero-size
# dimensions.  Here they exercise the same conversion helper (_vec) every entry point of this mirror goes through.
_SWIG_DEFAULT = -1          # swig_default_value


def _ivec(a, ndim, name="array"):
    arr = np.ascontiguousarray(np.asarray(a), dtype=np.int32)
    if arr.ndim != ndim:
        if arr.size == 0 and arr.ndim <= ndim:
            return arr.reshape((0,) * ndim)
        raise RuntimeError("%s must have %d dimensions, got %d" % (name, ndim, arr.ndim))
    return arr


def _seq_sum(a):
    total = np.float32(0)
    for v in a.ravel():
        total = np.float32(total + v)
    return float(total)


def test_vec_input(input):
    return _seq_sum(_vec(input, 1, "input"))


def test_ivec_input(input):
    return int(_ivec(input, 1, "input").sum())


def test_vec2_input(input):
    return _seq_sum(_vec(input, 2, "input"))


def test_vec3_input(input):
    return _seq_sum(_vec(input, 3, "input"))


def test_vec_output():
    return np.full(3, _SWIG_DEFAULT, np.float32)


def test_vec2_output():
    return np.full((3, 3), _SWIG_DEFAULT, np.float32)


def test_vec3_output():
    return np.full((3, 3, 3), _SWIG_DEFAULT, np.float32)


def test_ivec_output():
    return np.full(3, _SWIG_DEFAULT, np.int32)


def test_ivec2_output():
    return np.full((3, 3), _SWIG_DEFAULT, np.int32)


def test_ivec3_output():
    return np.full((3, 3, 3), _SWIG_DEFAULT, np.int32)


def test_vec_argout():
    return 0.0, np.full(10, _SWIG_DEFAULT, np.float32)


def test_vec2_argout():
    return 0.0, np.full((10, 10), _SWIG_DEFAULT, np.float32)


def test_array(v):
    return _vec(v, 1, "v")


def test_not_implemented_exception():
    raise RuntimeError("Not implemented")     # gridpp::not_implemented_exception -> RuntimeError (swig/gridpp.i:21-40)


for _f in (test_vec_input, test_ivec_input, test_vec2_input, test_vec3_input, test_vec_output, test_vec2_output, test_vec3_output, test_ivec_output,
           test_ivec2_output, test_ivec3_output, test_vec_argout, test_vec2_argout, test_array, test_not_implemented_exception):
    _f.__test__ = False       # not pytest tests
