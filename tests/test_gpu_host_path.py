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


@pytest.mark.parametrize("shape", [(1031, 1100), (1283, 1021), (2050, 640)])
@pytest.mark.parametrize("variance", [False, True], ids=["analysis", "analysis+variance"])
def test_banded_host_path_is_bit_identical_to_the_unbanded_one(shape, variance, monkeypatch):
    """Round 6: host arrays of a large grid go up, and the analysis comes down, in six bands of tile rows beside the first pass; the tiles the
    first pass leaves to the list passes are patched in by the host at the end.  Same bits as one upload / one download around the kernels
    (GPP_OI_NO_BANDS), with odd row and column counts (partial tiles, bands that do not divide the tile rows), on the first call of a
    geometry (read-back + serial list passes) and on the repeated one (list passes beside the first pass), with and without the variance,
    into the mirror's page-locked result arrays and into pageable ones (bands up only)."""
    import gridpp_amd as gridpp
    from oracle import oracle as O
    rng = np.random.default_rng(shape[0])
    Y, X = shape
    S = 4000     # dense enough for a few hundred declined tiles (unions beyond 40 rows where observations cluster)
    lats, lons = np.meshgrid(np.linspace(60, 61.5, Y), np.linspace(10, 13, X), indexing="ij")
    plat = np.concatenate([rng.uniform(60, 61.5, S - 600), rng.normal(60.7, 0.02, 600)])
    plon = np.concatenate([rng.uniform(10, 13, S - 600), rng.normal(11.3, 0.04, 600)])
    bg = rng.normal(0, 1, (Y, X)).astype(np.float32)
    bvar = rng.uniform(0.5, 2, (Y, X)).astype(np.float32)
    obs, pbg = rng.normal(0, 1, S).astype(np.float32), rng.normal(0, 1, S).astype(np.float32)
    ovar, bvp = rng.uniform(0.1, 1, S).astype(np.float32), rng.uniform(0.5, 2, S).astype(np.float32)
    grid, points, st = gridpp.Grid(lats, lons), gridpp.Points(plat, plon), gridpp.BarnesStructure(8000)

    def call():
        nonlocal bg, bvar, obs, ovar, pbg, bvp
        if variance:
            a, v = gridpp.optimal_interpolation_full(grid, bg, bvar, points, obs, ovar, pbg, bvp, st, 25)
            return np.array(a), np.array(v)
        return np.array(gridpp.optimal_interpolation(grid, bg, points, obs, ovar, pbg, st, 25)), None

    monkeypatch.setenv("GPP_OI_NO_BANDS", "1")
    ref_a, ref_v = call()
    declined = gridpp.oi_last_stats()["fallback_tiles"]
    assert declined > 20, declined     # (the patch must have something to do)
    monkeypatch.delenv("GPP_OI_NO_BANDS")
    for grid_fresh in (True, False, False):      # first call of a geometry, then the remembered-list path twice
        if grid_fresh:
            grid = gridpp.Grid(lats, lons)
        a, v = call()
        assert (a.view(np.uint32) == ref_a.view(np.uint32)).all(), int((a.view(np.uint32) != ref_a.view(np.uint32)).sum())
        if variance:
            assert (v.view(np.uint32) == ref_v.view(np.uint32)).all()
    monkeypatch.setenv("GPP_PAGEABLE_RESULTS", "1")   # (the mirror then hands out ordinary numpy arrays: bands up, one copy down)
    a, v = call()
    assert (a.view(np.uint32) == ref_a.view(np.uint32)).all()
    if variance:
        assert (v.view(np.uint32) == ref_v.view(np.uint32)).all()
    monkeypatch.delenv("GPP_PAGEABLE_RESULTS")
    # float64 arrays (numpy's default dtype): the bands go up as doubles and are cast on the upload stream -- the same float32 values, the same bits
    bg, bvar, obs, ovar, pbg, bvp = (x.astype(np.float64) for x in (bg, bvar, obs, ovar, pbg, bvp))
    a, v = call()
    assert a.dtype == np.float32 and (a.view(np.uint32) == ref_a.view(np.uint32)).all()
    if variance:
        assert (v.view(np.uint32) == ref_v.view(np.uint32)).all()
        return
    bg = bg.astype(np.float32)
    # and the unbanded result against the oracle on a sample of rows
    rows = rng.choice(Y, 6, replace=False)
    og = O.Pts(lats[rows].ravel(), lons[rows].ravel())
    ref = O.oi(og, bg[rows].ravel(), O.Pts(plat, plon), obs, ovar, pbg, O.Barnes(8000), 25).reshape(len(rows), X)
    err = np.abs(ref_a[rows] - ref) / np.maximum(np.abs(ref), 1e-3)
    assert err.max() < 1e-5, err.max()


@pytest.mark.parametrize("shape", [(1100, 1031), (4000, 300), (1027, 2050)])
def test_banded_neighbourhood_host_path_is_bit_identical(shape, monkeypatch):
    """Round 6: a large 2-D plane from numpy goes up in bands of the marching kernels' row segments beside their launches and the kernels write the page-locked
    result array themselves.  Same bits as one upload / the kernel / one download (GPP_NBH_NO_BANDS) for every marching statistic and halfwidths from 0 to the
    kernels' limits, with missing values, from float32 and float64 arrays, and into a pageable result array (no bands then)."""
    import gridpp_amd as gridpp
    rng = np.random.default_rng(shape[1])
    Y, X = shape
    f = rng.uniform(-5, 10, (Y, X)).astype(np.float32)
    f[rng.random((Y, X)) < 0.003] = np.nan
    f[Y // 2:Y // 2 + 40, X // 3:X // 3 + 50] = np.nan          # a block of missing values (the counted chunks of the box kernel)
    cases = [(gridpp.Mean, 0), (gridpp.Mean, 3), (gridpp.Mean, 16), (gridpp.Sum, 7), (gridpp.Count, 5), (gridpp.Min, 1), (gridpp.Max, 15), (gridpp.Max, 32), (gridpp.Min, 20)]
    for stat, hw in cases:
        monkeypatch.setenv("GPP_NBH_NO_BANDS", "1")
        ref = np.array(gridpp.neighbourhood(f, hw, stat))
        monkeypatch.delenv("GPP_NBH_NO_BANDS")
        for arr in (f, f.astype(np.float64)):
            out = np.array(gridpp.neighbourhood(arr, hw, stat))
            assert out.dtype == np.float32 and out.shape == ref.shape
            assert (out.view(np.uint32) == ref.view(np.uint32)).all(), (stat, hw, arr.dtype, int((out.view(np.uint32) != ref.view(np.uint32)).sum()))
    monkeypatch.setenv("GPP_PAGEABLE_RESULTS", "1")
    out = np.array(gridpp.neighbourhood(f, 7, gridpp.Mean))
    monkeypatch.setenv("GPP_NBH_NO_BANDS", "1")
    ref = np.array(gridpp.neighbourhood(f, 7, gridpp.Mean))
    assert (out.view(np.uint32) == ref.view(np.uint32)).all()
