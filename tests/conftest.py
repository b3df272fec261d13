import json
import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _nan(o):
    if o is None:
        return math.nan
    if isinstance(o, list):
        return [_nan(x) for x in o]
    if isinstance(o, dict):
        return {k: _nan(v) for k, v in o.items()}
    return o


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")) as f:
        return _nan(json.load(f))


# GRIDPP_TEST_POISON=1 (with tools/hostile/build.sh done): before EVERY call into the library the whole LDS of every CU and 500 registers per
# lane of every SIMD are filled with 0xFF (NaN as float and as double, -1 as an index), so that a kernel which reads something it never
# wrote -- LDS or registers left by whatever ran on the CU before -- fails here instead of once in a few thousand soak runs (round 4:
# that is how the intermittent failure of the max_points 33..62 soak was pinned down).
if os.environ.get("GRIDPP_TEST_POISON"):
    import ctypes as _C

    @pytest.fixture(scope="session", autouse=True)
    def _poison_every_library_call():
        import gridpp_amd
        lib = gridpp_amd._capi.lib()     # (first: it loads torch's HIP runtime, which the helper library then shares)
        plib = _C.CDLL(os.path.join(ROOT, "tools", "hostile", "libpoison.so"))
        skip = {"gpp_last_error", "gpp_version", "gpp_active_overrides", "gpp_oi_last_stats", "gpp_ensi_last_stats", "gpp_ensi_last_kernel_ms", "gpp_set_path_override", "gpp_ensi_set_convergence", "gpp_wait", "gpp_pending"}

        # with the -DGPP_POISON build of the library (GPP_LIB=gridpp_amd/lib/var_poison.so, tools/hostile/build.sh) every byte of every
        # call-to-call HBM workspace is 0xFF before each call as well (OI, EnSI + ensi_multi, neighbourhood; the remembered padding of
        # the quantile_fast byte planes alternately kept and forgotten)
        ws_poison = getattr(lib, "gpp_debug_poison_workspaces", None) if os.path.basename(gridpp_amd._capi.LIB_PATH) == "var_poison.so" else None
        if ws_poison is not None:
            ws_poison.argtypes = [_C.c_int, _C.c_int]
            ws_poison.restype = _C.c_int
        count = [0]

        class Poisoned:
            def __init__(self, fn):
                self.fn = fn

            def __call__(self, *a):
                assert plib.poison_lds(_C.c_uint(0xFFFFFFFF)) == 0 and plib.poison_regs(_C.c_uint(0xFFFFFFFF)) == 0
                if ws_poison is not None:
                    count[0] += 1
                    assert ws_poison(0xFF, count[0] & 1) == 0
                return self.fn(*a)

        for name in gridpp_amd._capi.SIGNATURES:
            if name not in skip and hasattr(lib, name):
                setattr(lib, name, Poisoned(getattr(lib, name)))
        yield


# The library reads no environment variable (round 4): its GPP_* path switches are set through gpp_set_path_override.  The tests keep
# writing them as environment variables (monkeypatch.setenv / os.environ): every call into the library first hands the CURRENT GPP_*
# variables of the process to that hook (and clears the ones that went away).
@pytest.fixture(scope="session", autouse=True)
def _path_overrides_follow_the_environment():
    try:
        import gridpp_amd
        lib = gridpp_amd._capi.lib()
    except Exception:      # noqa: BLE001  (no library built: the CPU-only tests that need none still run)
        yield
        return
    setter = lib.gpp_set_path_override
    state = {}

    def sync():
        now = {k: v for k, v in os.environ.items() if k.startswith("GPP_") and k != "GPP_LIB" and not k.startswith("GPP_BENCH_")}
        for k in list(state):
            if k not in now:
                setter(k.encode(), None)
                del state[k]
        for k, v in now.items():
            if state.get(k) != v:
                setter(k.encode(), v.encode())
                state[k] = v

    class Synced:
        def __init__(self, fn):
            self.fn = fn

        def __call__(self, *a):
            sync()
            return self.fn(*a)

    for name in gridpp_amd._capi.SIGNATURES:
        if name not in ("gpp_set_path_override", "gpp_last_error", "gpp_version") and hasattr(lib, name):
            setattr(lib, name, Synced(getattr(lib, name)))
    yield
