"""Parity at the FULL sizes of BASELINE.json's configs, where the oracle cannot run the whole problem: the HIP path is
checked on (a) a sample of cells against the oracle (cells are independent, so a sample is exact evidence for those cells),
(b) size-independent properties: tile independence (the same cells computed as a Points list give the same bits),
linearity in the innovations, pass-through where no observation is in range, window properties of the filters."""
import numpy as np
from tests.ensi_golden import rel_err
import pytest

pytestmark = pytest.mark.gpu


def _headline(ny=4000, nx=4000, S=10000):
    from bench import make_workload
    return make_workload(ny, nx, S, 1002, 0, ny)


def test_config3_oi_4000x4000_10k_obs():
    import gridpp_amd as gridpp
    from oracle import oracle as O
    lats, lons, bg, plat, plon, obs, ratios, pbg = _headline()
    grid = gridpp.Grid(lats, lons)
    points = gridpp.Points(plat, plon)
    st = gridpp.BarnesStructure(10000)
    out = gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, 30)
    assert out.shape == bg.shape and np.isfinite(out).all()
    assert np.abs(out - bg).max() > 0.5
    # (a) oracle on a sample of cells: 3 grid rows strided + 300 random cells
    rng = np.random.default_rng(0)
    op, ost = O.Pts(plat, plon), O.Barnes(10000)
    for r in (0, 1777, 3999):
        cols = np.arange(0, 4000, 40)
        ref = O.oi(O.Pts(lats[r, cols], lons[r, cols]), bg[r, cols], op, obs, ratios, pbg, ost, 30)
        err = np.abs(out[r, cols].astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-3)
        assert err.max() < 1e-5, (r, err.max())
    iy, ix = rng.integers(0, 4000, 300), rng.integers(0, 4000, 300)
    ref = O.oi(O.Pts(lats[iy, ix], lons[iy, ix]), bg[iy, ix], op, obs, ratios, pbg, ost, 30)
    err = np.abs(out[iy, ix].astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-3)
    assert err.max() < 1e-5
    # (b) tile independence: the same cells as a Points list (different tiling, different neighbours in the wave)
    sub = gridpp.optimal_interpolation(gridpp.Points(lats[iy, ix], lons[iy, ix]), bg[iy, ix], points, obs, ratios, pbg, st, 30)
    np.testing.assert_array_equal(sub, out[iy, ix])
    # (b) linearity in the innovations: with background == pbg == 0 the analysis is linear in obs
    z_g, z_p = np.zeros_like(bg), np.zeros_like(pbg)
    rows = slice(2000, 2008)
    g8 = gridpp.Grid(lats[rows], lons[rows])
    a1 = gridpp.optimal_interpolation(g8, z_g[rows], points, obs, ratios, z_p, st, 30)
    a2 = gridpp.optimal_interpolation(g8, z_g[rows], points, 2 * obs, ratios, z_p, st, 30)
    np.testing.assert_allclose(a2, 2 * a1, rtol=2e-6, atol=1e-6)


def test_config2_oi_1000x1000_1k_obs():
    import gridpp_amd as gridpp
    from oracle import oracle as O
    from bench import make_workload
    lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(1000, 1000, 1000, 1001, 0, 1000)
    out = gridpp.optimal_interpolation(gridpp.Grid(lats, lons), bg, gridpp.Points(plat, plon), obs, ratios, pbg,
                                       gridpp.BarnesStructure(10000), 20)
    rng = np.random.default_rng(1)
    iy, ix = rng.integers(0, 1000, 3000), rng.integers(0, 1000, 3000)
    ref = O.oi(O.Pts(lats[iy, ix], lons[iy, ix]), bg[iy, ix], O.Pts(plat, plon), obs, ratios, pbg, O.Barnes(10000), 20)
    np.testing.assert_array_equal(out[iy, ix], ref)   # bit-exact on every sampled cell


def test_config4_neighbourhood_4000x4000x100():
    import torch
    import gridpp_amd as gridpp
    from oracle import oracle as O
    Y = X = 4000
    E, hw = 100, 15
    g = torch.Generator(device="cuda").manual_seed(1003)
    cube = torch.rand((Y, X, E), generator=g, device="cuda") * 10
    mean = gridpp.neighbourhood(cube, hw, gridpp.Mean)
    thr = torch.linspace(0, 10, 11, device="cuda")
    qf = gridpp.neighbourhood_quantile_fast(cube, 0.5, hw, thr)
    assert mean.shape == (Y, X) and bool(torch.isfinite(mean).all()) and bool(torch.isfinite(qf).all())
    # (a) oracle on row bands: rows r-hw..r+hw of the cube give the exact answer for row r
    for r in (0, 2000, 3999):
        lo, hi = max(0, r - hw), min(Y, r + hw + 1)
        cols = slice(1000, 1200)
        band = cube[lo:hi, 1000 - hw:1200 + hw].cpu().numpy()
        ref = O.neighbourhood(band, hw, O.Mean)[r - lo, hw:hw + 200]
        got = mean[r, cols].cpu().numpy()
        assert (np.abs(got - ref) / np.abs(ref)).max() < 1e-5
        refq = O.neighbourhood_quantile_fast(band, [0.5], hw, thr.cpu().numpy())[r - lo, hw:hw + 200]
        gotq = qf[r, cols].cpu().numpy()
        assert (np.abs(gotq - refq) / np.maximum(np.abs(refq), 1e-3)).max() < 1e-5
    # (b) properties: a mean of U(0,10) members over 31x31x100 values is ~5; bounded by the data range
    assert 4.9 < float(mean[100:-100, 100:-100].mean()) < 5.1
    assert float(mean.min()) >= 0 and float(mean.max()) <= 10
    assert float(qf.min()) >= 0 and float(qf.max()) <= 10
    # (b) halfwidth 0 on a 2-D field is the identity
    f2 = cube[:, :, 0].contiguous()
    assert torch.equal(gridpp.neighbourhood(f2, 0, gridpp.Mean), f2)
    # (b) Min <= Mean <= Max pointwise
    mn, mx, me = gridpp.neighbourhood(f2, 7, gridpp.Min), gridpp.neighbourhood(f2, 7, gridpp.Max), gridpp.neighbourhood(f2, 7, gridpp.Mean)
    assert bool((mn <= me).all()) and bool((me <= mx).all())


def test_config5_ensi_2500x2500x50_sample():
    """Config 5 geometry (5 000 obs, 50 members, max_points 30) on a row band of the 2500x2500 grid + oracle sample."""
    import gridpp_amd as gridpp
    from oracle import oracle as O
    ny = nx = 2500
    E, S = 50, 5000
    rng = np.random.default_rng(1004)
    rows = np.array([0, 1249, 2499])
    lat1, lon1 = np.linspace(0, 1, ny)[rows], np.linspace(0, 1, nx)
    lats, lons = np.meshgrid(lat1, lon1, indexing="ij")
    bg = (np.sin(6 * lats) * np.cos(4 * lons) * 3)[:, :, None] + rng.normal(0, 1, (3, nx, E))
    bg = bg.astype(np.float32)
    plat, plon = rng.random(S), rng.random(S)
    pbg = rng.normal(0, 1, (S, E)).astype(np.float32)
    obs = rng.normal(0, 1, S).astype(np.float32)
    sig = np.ones(S, np.float32)
    out = gridpp.optimal_interpolation_ensi(gridpp.Grid(lats, lons), bg, gridpp.Points(plat, plon), obs, sig, pbg,
                                            gridpp.BarnesStructure(10000), 30)
    cols = np.arange(0, nx, 25)
    for k in range(3):
        ref = O.oi_ensi(O.Pts(lats[k, cols], lons[k, cols]), bg[k, cols], O.Pts(plat, plon), obs, sig, pbg, O.Barnes(10000), 30)
        err = rel_err(out[k, cols], ref, bg[k, cols])     # relative to max(|ref|, 1e-2, one float32 ulp of the cell's members)
        assert err.max() < 1e-5, err.max()
    assert np.abs(out - bg).max() > 0.1


def test_config5_ensi_full_grid_with_oracle_sample():
    """The FULL config 5 (2500 x 2500 x 50 members, 5 000 obs, max_points 30) in one call, inputs resident in HBM, checked against
    the oracle on 600 grid points: a regular lattice that includes the four corners and the edges, plus random points."""
    import torch
    import gridpp_amd as gridpp
    from oracle import oracle as O
    from tools.bench_cases import ensi_inputs
    ny = nx = 2500
    E, S = 50, 5000
    lats, lons, bg, plat, plon, pbg, obs, sig = ensi_inputs(ny, nx, E, S)
    out = gridpp.optimal_interpolation_ensi(gridpp.Grid(lats, lons), bg, gridpp.Points(plat, plon), obs, sig, pbg, gridpp.BarnesStructure(10000), 30)
    assert tuple(out.shape) == (ny, nx, E) and bool(torch.isfinite(out).all())
    rng = np.random.default_rng(77)
    yy, xx = np.meshgrid(np.linspace(0, ny - 1, 15).astype(int), np.linspace(0, nx - 1, 20).astype(int), indexing="ij")
    ys = np.concatenate([yy.ravel(), rng.integers(0, ny, 300)])
    xs = np.concatenate([xx.ravel(), rng.integers(0, nx, 300)])
    iy, ix = torch.from_numpy(ys).cuda(), torch.from_numpy(xs).cuda()
    got = out[iy, ix].cpu().numpy()
    bgs = bg[iy, ix].cpu().numpy()
    ref = O.oi_ensi(O.Pts(lats[ys, xs], lons[ys, xs]), bgs, O.Pts(plat, plon), obs.cpu().numpy(), sig.cpu().numpy(), pbg.cpu().numpy(),
                    O.Barnes(10000), 30)
    err = rel_err(got, ref, bgs)
    assert err.max() < 1e-5, err.max()
    plain = np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-2)      # north_star's plain measure: the default mode meets it since round 4
    assert plain.max() < 1e-5, plain.max()
    assert np.abs(got - bgs).max() > 0.1


def test_config5_ensi_full_grid_converged_mode_under_the_plain_measure():
    """Round-3 verdict, item 2: the FULL config 5 with the Jacobi sweeps run to convergence (gpp_ensi_set_convergence) against the
    oracle on 600 grid points under north_star's PLAIN measure |out - ref| / max(|ref|, 1e-2) < 1e-5 -- no ulp-aware floor."""
    import torch
    import gridpp_amd as gridpp
    from oracle import oracle as O
    from tools.bench_cases import ensi_inputs
    ny = nx = 2500
    E, S = 50, 5000
    lats, lons, bg, plat, plon, pbg, obs, sig = ensi_inputs(ny, nx, E, S)
    gridpp.ensi_set_convergence(True)
    try:
        out = gridpp.optimal_interpolation_ensi(gridpp.Grid(lats, lons), bg, gridpp.Points(plat, plon), obs, sig, pbg, gridpp.BarnesStructure(10000), 30)
    finally:
        gridpp.ensi_set_convergence(False)
    rng = np.random.default_rng(78)
    yy, xx = np.meshgrid(np.linspace(0, ny - 1, 15).astype(int), np.linspace(0, nx - 1, 20).astype(int), indexing="ij")
    ys = np.concatenate([yy.ravel(), rng.integers(0, ny, 300)])
    xs = np.concatenate([xx.ravel(), rng.integers(0, nx, 300)])
    iy, ix = torch.from_numpy(ys).cuda(), torch.from_numpy(xs).cuda()
    got = out[iy, ix].cpu().numpy().astype(np.float64)
    bgs = bg[iy, ix].cpu().numpy()
    ref = O.oi_ensi(O.Pts(lats[ys, xs], lons[ys, xs]), bgs, O.Pts(plat, plon), obs.cpu().numpy(), sig.cpu().numpy(), pbg.cpu().numpy(),
                    O.Barnes(10000), 30)
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-2)
    assert err.max() < 1e-5, err.max()


@pytest.mark.parametrize("elev", [True, "noise"], ids=["smooth_terrain", "white_noise_terrain"])
def test_config3_terrain_variants(elev, monkeypatch):
    """The two elevation / laf dependent variants of config 3 that bench.py times (BarnesStructure(10000, 200, 0.5) on smooth and on
    white-noise terrain) at the FULL 4000 x 4000 size: 600 sampled cells against the oracle -- on white noise every cell has its own
    observation set, k_oi scans and parks 16 M selections (2.1 GB) and k_oi_pairs solves them; on smooth terrain k_oi_union does 99 % and
    the lists the rest -- and, for the white-noise call, the same bits with GPP_OI_NO_PAIRS=1 (k_oi solving its selections itself)."""
    import gridpp_amd as gridpp
    from oracle import oracle as O
    from tools.bench_cases import oi_inputs
    ny = nx = 4000
    lats, lons, bg, plat, plon, obs, ratios, pbg, ge, gl, pe, pl, v, w = oi_inputs(ny, nx, 10000, 1002, elev)
    ge, gl = np.asarray(ge, np.float32), np.asarray(gl, np.float32)
    grid, points, st = gridpp.Grid(lats, lons, ge, gl), gridpp.Points(plat, plon, pe, pl), gridpp.BarnesStructure(10000, v, w)
    out = gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, 30)
    out = gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, 30)     # (the call bench.py times: the geometry remembers the first)
    stats = gridpp.oi_last_stats()
    assert out.shape == bg.shape and np.isfinite(out).all()
    rng = np.random.default_rng(3)
    yy, xx = np.meshgrid(np.linspace(0, ny - 1, 15).astype(int), np.linspace(0, nx - 1, 20).astype(int), indexing="ij")
    iy = np.concatenate([yy.ravel(), rng.integers(0, ny, 300)])
    ix = np.concatenate([xx.ravel(), rng.integers(0, nx, 300)])
    ref = O.oi(O.Pts(lats[iy, ix], lons[iy, ix], ge[iy, ix], gl[iy, ix]), bg[iy, ix], O.Pts(plat, plon, pe, pl), obs, ratios, pbg,
               O.Barnes(10000, v, w), 30)
    err = np.abs(out[iy, ix].astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-3)
    assert err.max() < 1e-5, err.max()
    if elev == "noise":
        assert stats["solves"] > 15_000_000 and stats["union_kernel_ms"] == 0      # one factorisation per cell, k_oi + k_oi_pairs alone
        monkeypatch.setenv("GPP_OI_NO_PAIRS", "1")
        out2 = gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, 30)
        np.testing.assert_array_equal(out2, out)
    else:
        assert stats["union_kernel_ms"] > 0 and stats["fallback_tiles"] < 0.05 * 250000


def test_spatially_varying_barnes_2000x2000_on_the_tile_path():
    """Round 6: the reference's "var len scale" benchmark shape with a length scale that really varies (+-20 % smoothly over ~50 km), 2000 x 2000 grid, 2 500
    observations, max_points 30 -- k_oi_union_sp (one unpivoted LU per tile).  (a) 400 sampled cells against the oracle's generic form with the scales at the
    first point of corr(p1, p2); (b) the pivoted LU per distinct selection (GPP_OI_NO_SP_UNION, round 5's path) within 1e-5 on every cell of a band of rows;
    (c) linearity in the innovations."""
    import os
    import gridpp_amd as gridpp
    from oracle import oracle as O
    lats, lons, bg, plat, plon, obs, ratios, pbg = _headline(2000, 2000, 2500)
    hf = (10000.0 * (1 + 0.2 * np.sin(12 * lats) * np.cos(9 * lons))).astype(np.float32)
    z = np.zeros_like(hf)
    grid, points = gridpp.Grid(lats, lons), gridpp.Points(plat, plon)
    st = gridpp.BarnesStructure(grid, hf, z, z)
    out = np.asarray(gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, 30))
    s = gridpp.oi_last_stats()
    assert s["union_kernel_ms"] > 0 and s["solves"] < 80000, s          # one factorisation per tile (62 500 tiles), not per distinct selection (~300 000)
    rng = np.random.default_rng(7)
    iy, ix = rng.integers(0, 2000, 400), rng.integers(0, 2000, 400)
    og, op = O.Pts(lats[iy, ix], lons[iy, ix]), O.Pts(plat, plon)
    min_rho = 0.0013
    loc = lambda h: np.array([O.structure_localization("Barnes", v, min_rho) for v in h], np.float32)
    # (the scales at the observations: the field at the nearest grid point -- the library's own `nearest`, bit-exact against the oracle in
    #  tests/test_gpu_nearest_parity.py; the oracle's brute force over 4 M x 2 500 pairs would take minutes)
    hc, ho = hf[iy, ix], np.asarray(gridpp.nearest(grid, points, hf)).astype(np.float32)
    zc, zo = np.zeros(400, np.float32), np.zeros(plat.size, np.float32)
    ref, _ = O.oi_full_generic(og, bg[iy, ix], np.ones(400, np.float32), op, obs, ratios, pbg, np.ones(plat.size, np.float32), O.Struct("Barnes", 10000.0), 30, True,
                               [hc, zc, zc, loc(hc)], [ho, zo, zo, loc(ho)])
    err = np.abs(out[iy, ix].astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-3)
    assert err.max() < 1e-5, err.max()
    rows = slice(992, 1008)
    g16 = gridpp.Grid(lats[rows], lons[rows])
    st16 = gridpp.BarnesStructure(g16, hf[rows], z[rows], z[rows])
    a_tile = np.asarray(gridpp.optimal_interpolation(g16, bg[rows], points, obs, ratios, pbg, st16, 30))
    os.environ["GPP_OI_NO_SP_UNION"] = "1"
    try:
        a_lu = np.asarray(gridpp.optimal_interpolation(g16, bg[rows], points, obs, ratios, pbg, st16, 30))
    finally:
        del os.environ["GPP_OI_NO_SP_UNION"]
    e2 = np.abs(a_tile.astype(np.float64) - a_lu) / np.maximum(np.abs(a_lu), 1e-3)
    assert e2.max() < 1e-5, e2.max()
    z_g, z_p = np.zeros_like(bg[rows]), np.zeros_like(pbg)
    a1 = np.asarray(gridpp.optimal_interpolation(g16, z_g, points, obs, ratios, z_p, st16, 30))
    a2 = np.asarray(gridpp.optimal_interpolation(g16, z_g, points, 2 * obs, ratios, z_p, st16, 30))
    np.testing.assert_allclose(a2, 2 * a1, rtol=2e-6, atol=1e-6)
