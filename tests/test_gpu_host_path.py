"""The host path (numpy arrays in, numpy array out) takes its device staging buffers from a pool (csrc/common.h: Staged, csrc/runtime.hip) since round 5:
fields of changing sizes in a row, buffers handed back and taken again by other fields, the pool emptied in between -- every result equal, bit for bit,
to the one the device path (torch tensors in HBM, no staging) gives for the same values."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_staging_pool_reuse_matches_the_device_path():
    import torch
    import gridpp_amd as gridpp
    rng = np.random.default_rng(42)
    sizes = [(300, 200), (40, 30), (300, 200), (700, 500), (64, 64), (700, 500), (10, 1000), (300, 200)]
    structure = gridpp.BarnesStructure(30000)
    for k, (Y, X) in enumerate(sizes):
        f = rng.uniform(0, 10, (Y, X)).astype(np.float32)
        f[rng.random((Y, X)) < 0.01] = np.nan
        d = torch.from_numpy(f).cuda()
        for hw, stat in ((3, gridpp.Mean), (20, gridpp.Mean), (7, gridpp.Max), (2, gridpp.Std)):
            host = np.asarray(gridpp.neighbourhood(f, hw, stat))
            dev = gridpp.neighbourhood(d, hw, stat).cpu().numpy()
            assert host.dtype == np.float32 and (host.view(np.uint32) == dev.view(np.uint32)).all(), (Y, X, hw, stat)
        thr = np.linspace(0, 10, 5).astype(np.float32)
        host = np.asarray(gridpp.neighbourhood_quantile_fast(f, 0.5, 3, thr))
        dev = gridpp.neighbourhood_quantile_fast(d, 0.5, 3, torch.from_numpy(thr).cuda()).cpu().numpy()
        assert (host.view(np.uint32) == dev.view(np.uint32)).all()
        # optimal interpolation: float32 and float64 host arrays (the latter are cast on the device) against device tensors
        lats, lons = np.meshgrid(np.linspace(60, 60.5, Y), np.linspace(10, 11, X), indexing="ij")
        S = 50
        plat, plon = rng.uniform(60, 60.5, S), rng.uniform(10, 11, S)
        obs, ratios, pbg = (rng.normal(0, 1, S).astype(np.float32), rng.uniform(0.1, 1, S).astype(np.float32), rng.normal(0, 1, S).astype(np.float32))
        grid, points = gridpp.Grid(lats, lons), gridpp.Points(plat, plon)
        bg = np.where(np.isnan(f), 0, f).astype(np.float32)
        host = np.asarray(gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, structure, 10))
        host64 = np.asarray(gridpp.optimal_interpolation(grid, bg.astype(np.float64), points, obs.astype(np.float64), ratios.astype(np.float64), pbg.astype(np.float64), structure, 10))
        dev = gridpp.optimal_interpolation(grid, torch.from_numpy(bg).cuda(), points, *(torch.from_numpy(a).cuda() for a in (obs, ratios, pbg)), structure, 10).cpu().numpy()
        assert (host.view(np.uint32) == dev.view(np.uint32)).all() and (host64.astype(np.float32).view(np.uint32) == dev.view(np.uint32)).all()
        if k in (2, 5):
            gridpp.release_workspaces()     # the pool is emptied; the next call allocates again
