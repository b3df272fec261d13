"""ctypes binding of libgridpp_hip.so (include/gridpp_hip.h).

The product path has NO CPU fallback: if the shared library is missing, or no
HIP device is visible when a compute entry point is called, this fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPP_LIB") or os.path.join(_HERE, "lib", "libgridpp_hip.so")   # GPP_LIB: A/B timing of two builds

GPP_OK, GPP_EINVAL, GPP_ERUNTIME, GPP_ENODEVICE = 0, -1, -2, -3
MEM_HOST, MEM_DEVICE, ASYNC, HOST_F64, Q_HOST = 0, 1, 2, 4, 8


class gpp_structure(C.Structure):
    _fields_ = [("kind", C.c_int), ("h", C.c_float), ("v", C.c_float), ("w", C.c_float), ("min_rho", C.c_float),
                ("kind_v", C.c_int), ("kind_w", C.c_int), ("loc", C.c_float), ("cv_dist", C.c_float), ("flags", C.c_int),
                ("field", C.c_void_p), ("field_v", C.c_void_p), ("field_w", C.c_void_p)]


class gpp_oi_stats(C.Structure):
    _fields_ = [("cells", C.c_longlong), ("cells_updated", C.c_longlong), ("solves", C.c_longlong),
                ("fallback_tiles", C.c_longlong), ("kernel_ms", C.c_float), ("union_kernel_ms", C.c_float), ("fallback_subtiles", C.c_longlong), ("big_cells", C.c_longlong)]


class gpp_ensi_stats(C.Structure):
    _fields_ = [("cells", C.c_longlong), ("condition_passthrough", C.c_longlong), ("real_part_passthrough", C.c_longlong), ("kernel_ms", C.c_float)]


_lib = None
fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int)
vp = C.c_void_p

# name -> (argtypes); every entry returns int except the two string getters
SIGNATURES = {
    "gpp_device_count": [ip],
    "gpp_set_device": [C.c_int],
    "gpp_get_stream": [C.POINTER(vp)],
    "gpp_synchronize": [],
    "gpp_host_alloc": [C.c_size_t, C.POINTER(vp)],
    "gpp_host_free": [vp],
    "gpp_points_create": [vp, vp, vp, vp, C.c_int, C.c_int, C.POINTER(vp)],
    "gpp_grid_create": [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)],
    "gpp_points_create_f64": [vp, vp, vp, vp, C.c_int, C.c_int, C.POINTER(vp)],
    "gpp_grid_create_f64": [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)],
    "gpp_points_destroy": [vp],
    "gpp_points_size": [vp, ip, ip, ip, ip],
    "gpp_points_get": [vp, C.c_int, vp],
    "gpp_convert_coordinates": [vp, vp, C.c_int, C.c_int, vp, vp, vp],
    "gpp_points_get_neighbours": [vp, C.c_float, C.c_float, C.c_float, C.c_int, vp, vp, C.c_int, ip],
    "gpp_points_get_closest_neighbours": [vp, C.c_float, C.c_float, C.c_int, C.c_int, vp, ip],
    "gpp_points_nearest_neighbour": [vp, vp, vp, C.c_int, C.c_int, vp],
    "gpp_nearest": [vp, vp, vp, vp, C.c_int],
    "gpp_nearest_levels": [vp, vp, vp, C.c_int, vp, C.c_int],
    "gpp_bilinear": [vp, vp, vp, C.c_int, vp, C.c_int],
    "gpp_fill": [vp, vp, vp, vp, C.c_float, C.c_int, vp, C.c_int],
    "gpp_fill_missing": [vp, C.c_int, C.c_int, vp, C.c_int],
    "gpp_doping": [vp, vp, vp, vp, vp, vp, C.c_float, vp, C.c_int],
    "gpp_neighbourhood_search": [vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, vp, vp, C.c_int],
    "gpp_calc_gradient": [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, vp, C.c_int],
    "gpp_points_get_neighbours_batch": [vp, vp, vp, C.c_int, C.c_float, C.c_int, vp, vp, vp, vp, C.c_longlong, C.POINTER(C.c_longlong)],
    "gpp_count": [vp, vp, C.c_float, vp, C.c_int],
    "gpp_staticcorr_points": [vp, vp, C.POINTER(gpp_structure), C.c_int, vp, C.c_int],
    "gpp_distance": [vp, vp, C.c_int, C.c_int, vp, C.c_int],
    "gpp_gridding": [vp, vp, vp, C.c_float, C.c_int, C.c_int, vp, C.c_int],
    "gpp_gridding_nearest": [vp, vp, vp, C.c_int, C.c_int, vp, C.c_int],
    "gpp_grid_get_box": [vp, vp, vp, C.c_int, vp, vp],
    "gpp_point_in_rectangle": [vp, C.c_float, C.c_float, ip],
    "gpp_structure_min_rho": [C.c_int, C.c_float, C.c_float, fp],
    "gpp_structure_localization_distance": [C.POINTER(gpp_structure), C.c_float, C.c_float, fp],
    "gpp_structure_corr": [C.POINTER(gpp_structure), fp, fp, C.c_int, fp],
    "gpp_field_create": [vp, vp, vp, vp, C.c_int, C.c_float, C.POINTER(vp)],
    "gpp_field_destroy": [vp],
    "gpp_optimal_interpolation_full": [vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(gpp_structure), C.c_int, C.c_int, vp, vp, C.c_int],
    "gpp_oi_last_stats": [C.POINTER(gpp_oi_stats)],
    "gpp_wait": [],
    "gpp_pending": [ip],
    "gpp_optimal_interpolation_ensi": [vp, vp, C.c_int, vp, vp, vp, vp, C.POINTER(gpp_structure), C.c_int, C.c_int, vp, C.c_int],
    "gpp_ensi_last_kernel_ms": [fp],
    "gpp_ensi_last_stats": [C.POINTER(gpp_ensi_stats)],
    "gpp_ensi_set_convergence": [C.c_int],
    "gpp_set_path_override": [C.c_char_p, C.c_char_p],
    "gpp_active_overrides": [C.c_char_p, C.c_int],
    "gpp_release_workspaces": [],
    "gpp_row_tile": [C.c_int, C.c_int, C.c_int, ip, ip],
    "gpp_comm_unique_id": [C.c_char_p],
    "gpp_comm_init": [C.c_int, C.c_int, C.c_char_p],
    "gpp_comm_rank": [ip, ip],
    "gpp_comm_broadcast": [vp, C.c_size_t, C.c_int],
    "gpp_comm_broadcast_host": [vp, C.c_size_t, C.c_int],
    "gpp_comm_halo_exchange": [vp, C.c_int, C.c_size_t, C.c_int, vp, ip],
    "gpp_comm_destroy": [],
    "gpp_optimal_interpolation_ensi_multi": [C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, C.POINTER(gpp_structure), C.c_int, C.c_int, vp, C.c_int],
    "gpp_calc_statistic": [vp, C.c_long, C.c_int, C.c_int, vp, C.c_int],
    "gpp_calc_quantile": [vp, C.c_long, C.c_int, vp, C.c_long, vp, C.c_int],
    "gpp_calc_even_quantiles": [vp, C.c_long, C.c_int, C.c_int, vp, ip, C.c_int],
    "gpp_neighbourhood": [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int],
    "gpp_neighbourhood_brute_force": [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp, C.c_int],
    "gpp_neighbourhood_quantile_fast": [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int],
}
STRING_GETTERS = ("gpp_last_error", "gpp_version")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("gridpp_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback)" % LIB_PATH)
        try:
            # torch bundles its own HIP runtime; load it first so that both share ONE runtime
            # (two copies of libamdhip64 in a process cannot both open the GPU)
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            f = getattr(L, name)
            f.argtypes = args
            f.restype = C.c_int
        for name in STRING_GETTERS:
            getattr(L, name).restype = C.c_char_p
            getattr(L, name).argtypes = []
        _lib = L
        # the library itself reads no environment variable; the GPP_* switches of THIS process' environment (tools/README.md: A/B runs,
        # `GPP_OI_NO_UNION=1 python tools/...`) are handed to its one override hook here, once
        for k, v in os.environ.items():
            if k.startswith("GPP_") and k not in ("GPP_LIB",) and not k.startswith("GPP_BENCH_"):
                L.gpp_set_path_override(k.encode(), v.encode())
    return _lib


def check(rc):
    """Error convention of swig/gridpp.i:21-40: invalid_argument -> ValueError, the rest -> RuntimeError."""
    if rc == GPP_OK:
        return
    msg = lib().gpp_last_error().decode("utf-8", "replace")
    if rc == GPP_EINVAL:
        raise ValueError(msg)
    raise RuntimeError(msg)
