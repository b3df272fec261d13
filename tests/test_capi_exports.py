"""CPU-only: libgridpp_hip.so loads without a GPU and exports every symbol include/gridpp_hip.h declares;
compute entry points fail loudly (GPP_ENODEVICE) instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gridpp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpp_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from gridpp_amd import _capi
    return _capi.lib()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from gridpp_amd import _capi
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "missing export: " + s
    bound = set(_capi.SIGNATURES) | set(_capi.STRING_GETTERS)
    assert set(syms) == bound, (set(syms) ^ bound)


def test_host_only_entry_points_work_without_gpu(lib):
    import gridpp_amd as gridpp
    assert gridpp.version().startswith("0.8.0")
    x, y, z = gridpp.convert_coordinates([0, 90], [0, 0])
    np.testing.assert_allclose(x, [6.378137e6, 0], atol=1)
    np.testing.assert_allclose(z, [0, 6.378137e6], atol=1)
    p = gridpp.Points([0, 1000, 2000], [0, 0, 0], [0, 0, 0], [0, 0, 0], gridpp.Cartesian)
    np.testing.assert_array_equal(gridpp.Points().get_closest_neighbours(0, 0, 5), [])
    with pytest.raises(ValueError):
        gridpp.Points([91], [0])
    g = gridpp.Grid(np.zeros((3, 4)), np.zeros((3, 4)))
    assert g.size() == [3, 4]
    assert gridpp.Grid().size() == [0, 0]


def test_compute_fails_loudly_without_gpu(lib):
    import gridpp_amd as gridpp
    if gridpp.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no HIP device"):
        gridpp.neighbourhood(np.ones((4, 4)), 1, gridpp.Mean)
    p3 = gridpp.Points([0, 1000, 2000], [0, 0, 0], [0, 0, 0], [0, 0, 0], gridpp.Cartesian)
    for query in (lambda: p3.get_neighbours(0, 0, 1001), lambda: p3.get_closest_neighbours(900, 0, 2),
                  lambda: gridpp.count(p3, p3, 1001), lambda: gridpp.gridding(p3, p3, [1, 2, 3], 1001, 0, gridpp.Mean),
                  lambda: gridpp.bilinear(gridpp.Grid([[0, 0], [1, 1]], [[0, 1], [0, 1]]), gridpp.Points([0.5], [0.5]), np.zeros((2, 2)))):
        with pytest.raises(RuntimeError, match="no HIP device"):
            query()
    pts = gridpp.Points([0], [0])
    with pytest.raises(RuntimeError, match="no HIP device"):
        gridpp.optimal_interpolation(pts, [0], pts, [1], [1], [0], gridpp.BarnesStructure(1000), 5)
