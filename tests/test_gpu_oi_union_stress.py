"""Randomised stress of the one-factorisation-per-tile path (k_oi_union) and its work lists against the CPU oracle:
varying grid shapes (partial tiles), observation densities (unions from a handful to far beyond the 40-row limit, so
that the 16-cell / 4-cell list passes and the k_oi remainder all run), clustered and duplicated observations (rho ties at
the max_points cut), missing values, max_points 1..32, anti-extrapolation, variance output, scattered output points
(tiles of unrelated cells)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def _check(out, ref):
    assert out.shape == ref.shape, (out.shape, ref.shape)
    assert (np.isnan(out) == np.isnan(ref)).all(), "NaN pattern differs in %d cells" % int((np.isnan(out) != np.isnan(ref)).sum())
    m = ~np.isnan(ref)
    err = np.abs(out[m].astype(np.float64) - ref[m]) / np.maximum(np.abs(ref[m]), 1e-3)
    assert err.max() < RTOL, err.max()


@pytest.mark.parametrize("seed", range(12))
def test_random_configurations(seed):
    _random_configuration(seed, [1, 2, 7, 20, 30, 32])


@pytest.mark.parametrize("seed", range(500, 512))
def test_random_configurations_one_factorisation_per_pair_of_tiles(seed, monkeypatch):
    """k_oi_union_pair (round 6, behind GPP_OI_PAIR_TILES: measured slower than one tile per wave): the two waves of a workgroup merge the unions
    of two neighbouring tiles and share one factorisation; pairs that do not fit, odd tile counts, tiles without cells to update and
    tiles the scan declines all go on as single tiles."""
    monkeypatch.setenv("GPP_OI_PAIR_TILES", "1")
    _random_configuration(seed, [1, 2, 7, 20, 30, 32])


def test_pair_of_tiles_halves_the_factorisations_on_a_regular_case(monkeypatch):
    import gridpp_amd as gridpp
    from oracle import oracle as O
    rng = np.random.default_rng(77)
    Y, X, S = 96, 128, 60     # (observations much sparser than tiles, as on the headline: neighbouring tiles select the same ones)
    lats, lons = np.meshgrid(np.linspace(60, 60.6, Y), np.linspace(10, 11.6, X), indexing="ij")
    plat, plon = 60 + 0.6 * rng.random(S), 10 + 1.6 * rng.random(S)
    bg = rng.normal(0, 1, (Y, X)).astype(np.float32)
    obs, pbg = rng.normal(0, 1, S).astype(np.float32), rng.normal(0, 1, S).astype(np.float32)
    ratios = rng.uniform(0.1, 1, S).astype(np.float32)
    grid, points, st = gridpp.Grid(lats, lons), gridpp.Points(plat, plon), gridpp.BarnesStructure(10000)
    single = gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, 10)
    n_single = gridpp.oi_last_stats()["solves"]
    monkeypatch.setenv("GPP_OI_PAIR_TILES", "1")
    paired = gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, 10)
    n_pair = gridpp.oi_last_stats()["solves"]
    assert n_pair < 0.7 * n_single, (n_pair, n_single)
    ref = O.oi(O.Pts(lats.ravel(), lons.ravel()), bg.ravel(), O.Pts(plat, plon), obs, ratios, pbg, O.Barnes(10000), 10).reshape(Y, X)
    _check(paired, ref)
    _check(single, ref)


@pytest.mark.parametrize("seed", range(100, 112))
def test_random_configurations_62_row_tile(seed):
    """max_points 33..62: the 64-column form of k_oi_union (two waves per workgroup) and k_oi<62> behind its work lists."""
    _random_configuration(seed, [33, 40, 50, 62], cressman=seed % 5 == 0)


@pytest.mark.parametrize("seed", range(700, 712))
def test_random_configurations_48_column_tile(seed, monkeypatch):
    """max_points 33..48 (round 6): the 48-column form of k_oi_union (48 register columns + 8 late columns, 56 slots, two waves per SIMD) -- at its upper end,
    where unions reach its 56 rows and the lists fill -- against the oracle, and the 64-column form (GPP_OI_NO_UNION48) on the same inputs."""
    _random_configuration(seed, [41, 44, 46, 47, 48], cressman=seed % 5 == 0)
    monkeypatch.setenv("GPP_OI_NO_UNION48", "1")
    _random_configuration(seed, [41, 44, 46, 47, 48], cressman=seed % 5 == 0)


@pytest.mark.parametrize("seed", range(200, 212))
def test_random_configurations_rough_terrain(seed):
    """White-noise elevations / land fractions with elevation and laf dependent rho: every cell has its own observation set, the
    per-selection kernel does everything (its per-lane candidate lists with the combined pruning test, its half-wave solves)."""
    import gridpp_amd as gridpp
    from oracle import oracle as O
    rng = np.random.default_rng(7000 + seed)
    Y, X = int(rng.integers(8, 60)), int(rng.integers(8, 60))
    S = int(rng.choice([40, 150, 600, 2000]))
    h = float(rng.choice([5000.0, 10000.0, 30000.0]))
    v = float(rng.choice([100.0, 300.0, 1000.0]))
    w = float(rng.choice([0.0, 0.5]))
    mp = int(rng.choice([3, 10, 30, 32, 50]))
    ext = 0.3 * float(rng.choice([0.3, 1.0]))
    lats, lons = np.meshgrid(np.linspace(60, 60 + ext, Y), np.linspace(10, 10 + 2 * ext, X), indexing="ij")
    ge, gl = rng.uniform(0, 1000, (Y, X)).astype(np.float32), rng.uniform(0, 1, (Y, X)).astype(np.float32)
    plat, plon = 60 + ext * rng.random(S), 10 + 2 * ext * rng.random(S)
    pe, pl = rng.uniform(0, 1000, S).astype(np.float32), rng.uniform(0, 1, S).astype(np.float32)
    if seed % 3 == 0:
        pe[rng.random(S) < 0.1] = np.nan     # observations without an elevation: their vertical factor is 1
    bg = rng.normal(0, 2, (Y, X)).astype(np.float32)
    obs, pbg = rng.normal(0, 2, S).astype(np.float32), rng.normal(0, 2, S).astype(np.float32)
    ratios = rng.uniform(0.05, 2, S).astype(np.float32)
    allow = bool(seed % 2)
    out = gridpp.optimal_interpolation(gridpp.Grid(lats, lons, ge, gl), bg, gridpp.Points(plat, plon, pe, pl), obs, ratios, pbg,
                                       gridpp.BarnesStructure(h, v, w), mp, allow)
    ref = O.oi(O.Pts(lats.ravel(), lons.ravel(), ge.ravel(), gl.ravel()), bg.ravel(), O.Pts(plat, plon, pe, pl), obs, ratios, pbg,
               O.Barnes(h, v, w), mp, allow).reshape(Y, X)
    _check(out, ref)


@pytest.mark.parametrize("seed", range(400, 410))
def test_parked_selections_rough_terrain(seed, monkeypatch):
    """The second call with a geometry whose tiles all declined goes to k_oi alone, which parks the selections of the cells that
    share their observation set with nobody (all of them here) for k_oi_pairs: analysis and analysis variance against the oracle,
    and the same bits as k_oi solving them itself (GPP_OI_NO_PAIRS)."""
    import gridpp_amd as gridpp
    from oracle import oracle as O
    rng = np.random.default_rng(9000 + seed)
    Y, X = int(rng.integers(8, 50)), int(rng.integers(8, 50))
    S = int(rng.choice([40, 150, 600, 2000]))
    h, v, w = float(rng.choice([5000.0, 10000.0, 30000.0])), float(rng.choice([100.0, 300.0])), float(rng.choice([0.0, 0.5]))
    mp = int(rng.choice([3, 10, 29, 30, 32]))
    ext = 0.3 * float(rng.choice([0.3, 1.0]))
    lats, lons = np.meshgrid(np.linspace(60, 60 + ext, Y), np.linspace(10, 10 + 2 * ext, X), indexing="ij")
    ge, gl = rng.uniform(0, 1000, (Y, X)).astype(np.float32), rng.uniform(0, 1, (Y, X)).astype(np.float32)
    plat, plon = 60 + ext * rng.random(S), 10 + 2 * ext * rng.random(S)
    pe, pl = rng.uniform(0, 1000, S).astype(np.float32), rng.uniform(0, 1, S).astype(np.float32)
    bg = rng.normal(0, 2, (Y, X)).astype(np.float32)
    bg[rng.random((Y, X)) < 0.02] = np.nan                 # cells without a background: untouched, and never parked
    bvar = rng.uniform(0.5, 2, (Y, X)).astype(np.float32)
    obs, pbg = rng.normal(0, 2, S).astype(np.float32), rng.normal(0, 2, S).astype(np.float32)
    ovar, pbvar = rng.uniform(0.05, 2, S).astype(np.float32), rng.uniform(0.5, 2, S).astype(np.float32)
    allow = bool(seed % 2)
    grid, points, st = gridpp.Grid(lats, lons, ge, gl), gridpp.Points(plat, plon, pe, pl), gridpp.BarnesStructure(h, v, w)
    call = lambda: gridpp.optimal_interpolation_full(grid, bg, bvar, points, obs, ovar, pbg, pbvar, st, mp, allow)
    call()                                                  # (first call: the first pass of k_oi_union declines, lists follow)
    out, var = call()
    stats = gridpp.oi_last_stats()
    if stats["union_kernel_ms"] != 0 or stats["solves"] == 0:
        pytest.skip("the first pass keeps more than half of the tiles of this geometry (few observations): k_oi never runs alone")
    ref, rvar = O.oi_full(O.Pts(lats.ravel(), lons.ravel(), ge.ravel(), gl.ravel()), bg.ravel(), bvar.ravel(), O.Pts(plat, plon, pe, pl), obs,
                          ovar, pbg, pbvar, O.Barnes(h, v, w), mp, allow)
    _check(out, ref.reshape(Y, X))
    _check(var, rvar.reshape(Y, X))
    if seed % 2:
        gridpp.release_workspaces()                         # the parked selections are given back and come again on demand
        out3, var3 = call()
        assert np.array_equal(out, out3, equal_nan=True) and np.array_equal(var, var3, equal_nan=True)
    monkeypatch.setenv("GPP_OI_NO_PAIRS", "1")
    out2, var2 = call()
    assert gridpp.oi_last_stats()["solves"] == stats["solves"]
    assert np.array_equal(out, out2, equal_nan=True) and np.array_equal(var, var2, equal_nan=True)


def test_parked_selections_behind_a_long_work_list():
    """First call on rough terrain with enough tiles for the two-level list passes: what they leave to k_oi (everything) is parked
    for k_oi_pairs as well, the counts of the cells k_oi does not visit cleared first."""
    import gridpp_amd as gridpp
    from oracle import oracle as O
    rng = np.random.default_rng(77)
    Y, X, S = 150, 170, 900
    lats, lons = np.meshgrid(np.linspace(60, 60.6, Y), np.linspace(10, 11.2, X), indexing="ij")
    ge, gl = rng.uniform(0, 1000, (Y, X)).astype(np.float32), rng.uniform(0, 1, (Y, X)).astype(np.float32)
    ge[:40, :40] = 100.0; gl[:40, :40] = 0.5                # a smooth corner: its tiles are solved by k_oi_union, never visited by k_oi
    plat, plon = 60 + 0.6 * rng.random(S), 10 + 1.2 * rng.random(S)
    pe, pl = rng.uniform(0, 1000, S).astype(np.float32), rng.uniform(0, 1, S).astype(np.float32)
    bg = rng.normal(0, 2, (Y, X)).astype(np.float32)
    obs, pbg = rng.normal(0, 2, S).astype(np.float32), rng.normal(0, 2, S).astype(np.float32)
    ratios = rng.uniform(0.05, 2, S).astype(np.float32)
    ref = O.oi(O.Pts(lats.ravel(), lons.ravel(), ge.ravel(), gl.ravel()), bg.ravel(), O.Pts(plat, plon, pe, pl), obs, ratios, pbg,
               O.Barnes(8000.0, 200.0, 0.5), 30).reshape(Y, X)
    points, st = gridpp.Points(plat, plon, pe, pl), gridpp.BarnesStructure(8000.0, 200.0, 0.5)
    for _ in range(2):                                       # (a call on a large smooth grid in between leaves stale counts behind)
        out = gridpp.optimal_interpolation(gridpp.Grid(lats, lons, ge, gl), bg, points, obs, ratios, pbg, st, 30)
        stats = gridpp.oi_last_stats()
        assert stats["union_kernel_ms"] > 0 and stats["fallback_tiles"] > 100
        _check(out, ref)
        gl2 = rng.uniform(0, 1, (Y + 20, X + 20)).astype(np.float32)
        la2, lo2 = np.meshgrid(np.linspace(60, 60.6, Y + 20), np.linspace(10, 11.2, X + 20), indexing="ij")
        g2 = gridpp.Grid(la2, lo2, rng.uniform(0, 1000, (Y + 20, X + 20)).astype(np.float32), gl2)
        b2 = rng.normal(0, 2, (Y + 20, X + 20)).astype(np.float32)
        gridpp.optimal_interpolation(g2, b2, points, obs, ratios, pbg, st, 30)
        gridpp.optimal_interpolation(g2, b2, points, obs, ratios, pbg, st, 30)    # direct path: every count of the larger grid written


def test_long_list_longer_than_remembered_is_run_again_without_counting_twice(monkeypatch):
    """A geometry that remembers a long work list launches the two-level list passes for the remembered length without asking the host
    (csrc/oi.hip, `expect_long`); when MORE tiles are declined than those grids hold, the passes run again with the true length.  Forced
    here with GPP_OI_LONG_CAP: same bits as the first call, and the statistics of the call count every cell once (the
    atomics of both runs used to add up)."""
    import gridpp_amd as gridpp
    rng = np.random.default_rng(79)
    Y, X, S = 640, 240, 800       # (sparse observations: the 64 cells of a smooth tile select almost the same 30)
    lats, lons = np.meshgrid(np.linspace(60, 60 + Y / 240.0, Y), np.linspace(10, 12, X), indexing="ij")
    ge, gl = rng.uniform(0, 1000, (Y, X)).astype(np.float32), rng.uniform(0, 1, (Y, X)).astype(np.float32)
    ge[:560] = 100.0; gl[:560] = 0.5                          # smooth: k_oi_union keeps these tiles; the rough rest is declined (less than half)
    plat, plon = 60 + Y / 240.0 * rng.random(S), 10 + 2 * rng.random(S)
    pe, pl = rng.uniform(0, 1000, S).astype(np.float32), rng.uniform(0, 1, S).astype(np.float32)
    pe[plat < 60 + 560 / 240.0] = 100.0; pl[plat < 60 + 560 / 240.0] = 0.5      # (the observations over the smooth rows as well)
    bg = rng.normal(0, 2, (Y, X)).astype(np.float32)
    obs, pbg = rng.normal(0, 2, S).astype(np.float32), rng.normal(0, 2, S).astype(np.float32)
    ratios = rng.uniform(0.05, 2, S).astype(np.float32)
    grid, points, st = gridpp.Grid(lats, lons, ge, gl), gridpp.Points(plat, plon, pe, pl), gridpp.BarnesStructure(8000.0, 200.0, 0.5)
    out1 = np.asarray(gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, 30))
    s1 = gridpp.oi_last_stats()
    assert s1["union_kernel_ms"] > 0 and 192 < s1["fallback_tiles"] < 0.5 * (Y * X / 64)      # a long list, and the first pass still pays
    out2 = np.asarray(gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, 30))   # (remembered length: one run)
    s2 = gridpp.oi_last_stats()
    monkeypatch.setenv("GPP_OI_LONG_CAP", "8")
    out3 = np.asarray(gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, 30))   # (grids for 8 tiles: run again)
    s3 = gridpp.oi_last_stats()
    assert np.array_equal(out1, out2, equal_nan=True) and np.array_equal(out1, out3, equal_nan=True)
    for k in ("cells", "cells_updated", "solves", "fallback_tiles", "fallback_subtiles"):
        assert s1[k] == s2[k] == s3[k], (k, s1[k], s2[k], s3[k])


@pytest.mark.parametrize("long_list", [False, True], ids=["short_list", "long_list"])
def test_list_passes_beside_the_first_pass_from_the_remembered_list(long_list, monkeypatch):
    """Round 5 (csrc/oi.hip, `overlap`): a geometry keeps the list of the tiles its first pass declined; the next call with the same Grid
    handle runs the list passes over the REMEMBERED list on a second stream while the first pass skips those tiles.  Nothing may depend
    on the memory being right: (1) the second and third call equal the first bit for bit and report the same statistics; (2) with other
    observations unusable (the first pass declines OTHER tiles: new ones go through the serial passes and join the memory, remembered
    ones that would fit now are solved by the list passes all the same) the result equals that of a fresh handle; (3) back to the first
    inputs; (4) the same with the overlap switched off.  /root/reference/src/api/oi.cpp:221-338 has no state between grid points."""
    import gridpp_amd as gridpp
    rng = np.random.default_rng(81 + long_list)
    Y, X, S = (640, 240, 800) if long_list else (200, 120, 130)       # (sparse observations: the 64 cells of a smooth tile select almost the same 30)
    lats, lons = np.meshgrid(np.linspace(60, 60 + Y / 240.0, Y), np.linspace(10, 10 + X / 120.0, X), indexing="ij")
    ge, gl = rng.uniform(0, 1000, (Y, X)).astype(np.float32), rng.uniform(0, 1, (Y, X)).astype(np.float32)
    smooth = 560 if long_list else 190
    ge[:smooth] = 100.0; gl[:smooth] = 0.5                     # smooth rows: the first pass keeps them; the rough rest is declined
    plat, plon = 60 + Y / 240.0 * rng.random(S), 10 + X / 120.0 * rng.random(S)
    pe, pl = rng.uniform(0, 1000, S).astype(np.float32), rng.uniform(0, 1, S).astype(np.float32)
    pe[plat < 60 + smooth / 240.0] = 100.0; pl[plat < 60 + smooth / 240.0] = 0.5
    bg = rng.normal(0, 2, (Y, X)).astype(np.float32)
    obs, pbg = rng.normal(0, 2, S).astype(np.float32), rng.normal(0, 2, S).astype(np.float32)
    obs2 = obs.copy(); obs2[rng.random(S) < 0.3] = np.nan     # other usable observations: other selections, other declined tiles
    ratios = rng.uniform(0.05, 2, S).astype(np.float32)
    points, st = gridpp.Points(plat, plon, pe, pl), gridpp.BarnesStructure(8000.0, 200.0, 0.5)

    def fresh(o):
        out = np.asarray(gridpp.optimal_interpolation(gridpp.Grid(lats, lons, ge, gl), bg, points, o, ratios, pbg, st, 30))
        return out, gridpp.oi_last_stats()
    ref1, sr1 = fresh(obs)
    ref2, sr2 = fresh(obs2)
    assert sr1["union_kernel_ms"] > 0 and sr1["fallback_tiles"] > (192 if long_list else 0) and 16 * sr1["fallback_tiles"] > (3072 if long_list else 0)
    if not long_list:
        assert 16 * sr1["fallback_tiles"] <= 3072
    grid = gridpp.Grid(lats, lons, ge, gl)
    keys = ("cells", "cells_updated", "solves")
    for step, (o, ref, sr) in enumerate([(obs, ref1, sr1), (obs, ref1, sr1), (obs, ref1, sr1), (obs2, ref2, sr2), (obs2, ref2, sr2), (obs, ref1, sr1), (obs, ref1, sr1)]):
        out = np.asarray(gridpp.optimal_interpolation(grid, bg, points, o, ratios, pbg, st, 30))
        s = gridpp.oi_last_stats()
        assert np.array_equal(out, ref, equal_nan=True), step
        assert s["cells"] == sr["cells"] and s["cells_updated"] == sr["cells_updated"], (step, s, sr)
        # (the memory only grows: remembered tiles stay listed -- until more than half of the tiles are, and the call goes to k_oi alone)
        assert s["fallback_tiles"] >= sr["fallback_tiles"] or s["union_kernel_ms"] == 0, (step, s, sr)
    monkeypatch.setenv("GPP_OI_NO_OVERLAP", "1")
    out = np.asarray(gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, 30))
    s = gridpp.oi_last_stats()
    assert np.array_equal(out, ref1, equal_nan=True)
    if s["union_kernel_ms"] > 0:       # (not when the memory holds more than half of the tiles by now: that call went to k_oi alone)
        for k in keys + ("fallback_tiles",):
            assert s[k] == sr1[k], (k, s[k], sr1[k])


@pytest.mark.parametrize("seed", range(300, 312))
def test_random_configurations_pivoted_lu(seed, monkeypatch):
    """The pivoted-LU form of k_oi (non-symmetric / spatially varying structures; forced here on symmetric systems, whose oracle
    values are known): one solve per distinct selection and a dot product per cell without a variance output, a pair of substitutions
    per cell with one -- on the same random configurations, max_points up to 62."""
    monkeypatch.setenv("GPP_OI_FORCE_LU", "1")
    _random_configuration(seed, [1, 5, 20, 30, 32, 45, 62], lu=True)


def _random_inputs(seed, mps):
    """The inputs of one random configuration (shared with tools/oi_hostile_soak.py, which caches the oracle's answers)."""
    rng = np.random.default_rng(4000 + seed)
    Y, X = int(rng.integers(5, 70)), int(rng.integers(5, 70))
    S = int(rng.choice([3, 12, 40, 150, 600, 2500]))
    h = float(rng.choice([3000.0, 10000.0, 40000.0]))
    mp = int(rng.choice(mps))
    ext = 0.3 * float(rng.choice([0.2, 1.0, 3.0]))          # domain size in degrees: tile size relative to h varies
    lats, lons = np.meshgrid(np.linspace(60, 60 + ext, Y), np.linspace(10, 10 + 2 * ext, X), indexing="ij")
    kind = seed % 3
    if kind == 0:      # uniform
        plat, plon = 60 + ext * rng.random(S), 10 + 2 * ext * rng.random(S)
    elif kind == 1:    # clustered: a few dense blobs, many near-coincident observations
        cy, cx = 60 + ext * rng.random(4), 10 + 2 * ext * rng.random(4)
        k = rng.integers(0, 4, S)
        plat, plon = cy[k] + 0.01 * ext * rng.standard_normal(S), cx[k] + 0.02 * ext * rng.standard_normal(S)
    else:              # exact duplicates: equal rho for several observations at the cut
        base = max(1, S // 3)
        by, bx = 60 + ext * rng.random(base), 10 + 2 * ext * rng.random(base)
        k = rng.integers(0, base, S)
        plat, plon = by[k], bx[k]
    bg = rng.normal(0, 2, (Y, X)).astype(np.float32)
    obs = rng.normal(0, 2, S).astype(np.float32)
    pbg = rng.normal(0, 2, S).astype(np.float32)
    ratios = rng.uniform(0.05, 2, S).astype(np.float32)
    if seed % 4 == 1:
        obs[rng.random(S) < 0.1] = np.nan
        bg[rng.random((Y, X)) < 0.05] = np.nan
    allow = bool(seed % 2)
    # (drawn last, as the variance part of the check always did)
    bvar = rng.uniform(0.5, 2, (Y, X)).astype(np.float32)
    bvp = rng.uniform(0.5, 2, S).astype(np.float32)
    return dict(Y=Y, X=X, S=S, h=h, mp=mp, lats=lats, lons=lons, plat=plat, plon=plon, bg=bg, obs=obs, pbg=pbg, ratios=ratios,
                allow=allow, bvar=bvar, bvp=bvp)


def _random_configuration(seed, mps, cressman=False, lu=False):
    import gridpp_amd as gridpp
    from oracle import oracle as O
    c = _random_inputs(seed, mps)
    Y, X, S, h, mp, allow = c["Y"], c["X"], c["S"], c["h"], c["mp"], c["allow"]
    lats, lons, plat, plon, bg, obs, pbg, ratios = c["lats"], c["lons"], c["plat"], c["plon"], c["bg"], c["obs"], c["pbg"], c["ratios"]
    grid, points, st = gridpp.Grid(lats, lons), gridpp.Points(plat, plon), gridpp.BarnesStructure(h)
    og, op, ost = O.Pts(lats.ravel(), lons.ravel()), O.Pts(plat, plon), O.Barnes(h)
    if cressman:    # a symmetric structure function that is not the Barnes fast path (k_oi_union<false, ...>)
        st = gridpp.CressmanStructure(h)
        ones_g, ones_p = np.ones(Y * X, np.float32), np.ones(S, np.float32)
        out = gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, mp, allow)
        ref, _ = O.oi_full_generic(og, bg.ravel(), ones_g, op, obs, ratios, pbg, ones_p, O.Struct("Cressman", h), mp, allow)
        _check(out, ref.reshape(Y, X))
        assert gridpp.oi_last_stats()["union_kernel_ms"] > 0
        return
    out = gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, mp, allow)
    ref = O.oi(og, bg.ravel(), op, obs, ratios, pbg, ost, mp, allow).reshape(Y, X)
    _check(out, ref)
    stats = gridpp.oi_last_stats()
    assert (stats["union_kernel_ms"] > 0) != lu, stats  # this configuration is routed to k_oi_union (unless the pivoted LU is forced)
    # variance output of the same configuration
    bvar, bvp = c["bvar"], c["bvp"]
    out2, var = gridpp.optimal_interpolation_full(grid, bg, bvar, points, obs, ratios, pbg, bvp, st, mp, allow)
    ref2, rvar = O.oi_full(og, bg.ravel(), bvar.ravel(), op, obs, ratios, pbg, bvp, ost, mp, allow)
    _check(out2, ref2.reshape(Y, X))
    _check(var, rvar.reshape(Y, X))


def _poison_chip():
    """0xFF over every byte of LDS and 500 registers per lane (tools/hostile/poison.hip, built by build()); False when the helper is absent"""
    import ctypes as C
    import os
    import gridpp_amd
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "hostile", "libpoison.so")
    if not os.path.exists(so):
        return False
    gridpp_amd._capi.lib()            # (first: the helper then shares the HIP runtime torch brought)
    plib = C.CDLL(so)
    assert plib.poison_lds(C.c_uint(0xFFFFFFFF)) == 0 and plib.poison_regs(C.c_uint(0xFFFFFFFF)) == 0
    return True


def test_62_row_sequence_in_one_process_after_an_unrelated_call():
    """Round-3 verdict, item 1c: the sequence that failed once at seed 5019 (max_points 33..62: the 64-column k_oi_union and k_oi<62> behind
    its lists), in one process, behind a large unrelated call -- every seed against the oracle."""
    import gridpp_amd as gridpp
    rng = np.random.default_rng(5)
    Y, X, S = 700, 900, 4000
    lats, lons = np.meshgrid(np.linspace(60, 62, Y), np.linspace(10, 14, X), indexing="ij")
    v = rng.normal(0, 1, S).astype(np.float32)
    gridpp.optimal_interpolation(gridpp.Grid(lats, lons), rng.normal(0, 1, (Y, X)).astype(np.float32), gridpp.Points(60 + 2 * rng.random(S), 10 + 4 * rng.random(S)),
                                 v, np.abs(v) + 0.1, v, gridpp.BarnesStructure(12000.0), 25)
    for seed in range(5000, 5041):
        _random_configuration(seed, [33, 40, 50, 62])


def test_one_extras_row_behind_a_61_row_core_reads_no_stale_lds():
    """The root cause of that failure: k_oi_union's per-cell finish reads the rows of B four columns at a time, up to
    three doubles past a row; with ONE extras row behind a core of c = 4 k + 1 rows the third lies behind d' -- for c >= 57 (the 64-column
    form) outside the 16 KB of rho slots the kernel initialises, i.e. in LDS left by whatever workgroup ran on the CU before, and
    0 x NaN = NaN.  Seed 5019 (max_points 62: 61 core rows + one extra in a 4-cell item) hits it on its first call; with every byte of
    LDS set to 0xFF beforehand the failure was deterministic (profiles/r04_oi_hostile_soak_before_fix.txt)."""
    _poison_chip()
    _random_configuration(5019, [33, 40, 50, 62])
    _poison_chip()
    _random_configuration(5019, [33, 40, 50, 62])


def test_scattered_output_points_use_the_work_lists():
    """Output points in random order: the 64 cells of a tile are unrelated, their union is far above 40 rows, so the
    tiles go down the 16-cell and 4-cell lists and what is left to k_oi -- and every path must give the same values."""
    import gridpp_amd as gridpp
    from oracle import oracle as O
    rng = np.random.default_rng(77)
    C, S = 5000, 3000
    qlat, qlon = 60 + rng.random(C), 10 + 2 * rng.random(C)
    plat, plon = 60 + rng.random(S), 10 + 2 * rng.random(S)
    bg = rng.normal(0, 1, C).astype(np.float32)
    obs, pbg = rng.normal(0, 1, S).astype(np.float32), rng.normal(0, 1, S).astype(np.float32)
    ratios = rng.uniform(0.1, 1, S).astype(np.float32)
    st = gridpp.BarnesStructure(8000)
    out = gridpp.optimal_interpolation(gridpp.Points(qlat, qlon), bg, gridpp.Points(plat, plon), obs, ratios, pbg, st, 12)
    stats = gridpp.oi_last_stats()
    assert stats["fallback_tiles"] > 0
    ref = O.oi(O.Pts(qlat, qlon), bg, O.Pts(plat, plon), obs, ratios, pbg, O.Barnes(8000), 12)
    _check(out, ref)
    # the same cells sorted along a space-filling order (coherent tiles): identical values, bit for bit
    order = np.lexsort((np.floor(qlon * 40), np.floor(qlat * 40)))
    out_s = gridpp.optimal_interpolation(gridpp.Points(qlat[order], qlon[order]), bg[order], gridpp.Points(plat, plon), obs, ratios,
                                         pbg, st, 12)
    _check(out_s, ref[order])


def test_repeated_calls_when_the_last_call_misleads_the_next():
    """A geometry remembers what its last call did (tiles declined by the first pass, items left to k_oi) and launches the list
    passes accordingly, without asking the device first.  Here the memory is wrong on purpose: the first call has no valid
    background anywhere (nothing declined, nothing left), the second one the real field (declined tiles, items for k_oi), the
    third the real field again (now expected), the fourth nothing again -- all must give the oracle's values."""
    import gridpp_amd as gridpp
    from oracle import oracle as O
    rng = np.random.default_rng(78)
    C, S = 5000, 3000
    qlat, qlon = 60 + rng.random(C), 10 + 2 * rng.random(C)
    plat, plon = 60 + rng.random(S), 10 + 2 * rng.random(S)
    bg = rng.normal(0, 1, C).astype(np.float32)
    obs, pbg = rng.normal(0, 1, S).astype(np.float32), rng.normal(0, 1, S).astype(np.float32)
    ratios = rng.uniform(0.1, 1, S).astype(np.float32)
    st, q, p = gridpp.BarnesStructure(8000), gridpp.Points(qlat, qlon), gridpp.Points(plat, plon)
    nothing = np.full(C, np.nan, np.float32)
    out0 = gridpp.optimal_interpolation(q, nothing, p, obs, ratios, pbg, st, 30)
    assert np.isnan(np.asarray(out0)).all()
    s0 = gridpp.oi_last_stats()
    assert s0["fallback_tiles"] == 0 and s0["fallback_subtiles"] == 0
    ref = O.oi(O.Pts(qlat, qlon), bg, O.Pts(plat, plon), obs, ratios, pbg, O.Barnes(8000), 30)
    out = gridpp.optimal_interpolation(q, bg, p, obs, ratios, pbg, st, 30)
    s1 = gridpp.oi_last_stats()
    assert s1["fallback_tiles"] > 0 and s1["fallback_subtiles"] > 0   # (neither was expected: both lists were found by the read-back)
    _check(out, ref)
    out = gridpp.optimal_interpolation(q, bg, p, obs, ratios, pbg, st, 30)   # (now every tile is expected to decline: k_oi alone)
    _check(out, ref)
    # and back: a call that expects work and finds none
    out0 = gridpp.optimal_interpolation(q, nothing, p, obs, ratios, pbg, st, 30)
    assert np.isnan(np.asarray(out0)).all()


def test_union_and_per_selection_kernels_agree_on_the_headline_geometry():
    import os
    import gridpp_amd as gridpp
    from bench import make_workload
    lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(400, 400, 1000, 1002, 0, 400)
    grid, points, st = gridpp.Grid(lats, lons), gridpp.Points(plat, plon), gridpp.BarnesStructure(10000)
    a = gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, 30)
    assert gridpp.oi_last_stats()["union_kernel_ms"] > 0
    os.environ["GPP_OI_NO_UNION"] = "1"
    try:
        b = gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, 30)
        assert gridpp.oi_last_stats()["union_kernel_ms"] == 0
    finally:
        del os.environ["GPP_OI_NO_UNION"]
    err = np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b), 1e-3)
    assert err.max() < 2e-6


@pytest.mark.parametrize("shape", [(12, 900), (900, 12), (300, 40)])
def test_anisotropic_grids_pick_a_matching_tile_shape(shape):
    """Cells far from square (the 64-cell tile becomes 1x64 ... 64x1): same values as the oracle whatever the tile shape."""
    import os
    import gridpp_amd as gridpp
    from oracle import oracle as O
    Y, X = shape
    rng = np.random.default_rng(Y * 1000 + X)
    lats, lons = np.meshgrid(np.linspace(60, 60.5, Y), np.linspace(10, 11, X), indexing="ij")
    S = 700
    plat, plon = 60 + 0.5 * rng.random(S), 10 + rng.random(S)
    bg = rng.normal(0, 1, (Y, X)).astype(np.float32)
    obs, pbg = rng.normal(0, 1, S).astype(np.float32), rng.normal(0, 1, S).astype(np.float32)
    ratios = rng.uniform(0.1, 1, S).astype(np.float32)
    grid, points, st = gridpp.Grid(lats, lons), gridpp.Points(plat, plon), gridpp.BarnesStructure(6000)
    ref = O.oi(O.Pts(lats.ravel(), lons.ravel()), bg.ravel(), O.Pts(plat, plon), obs, ratios, pbg, O.Barnes(6000), 25).reshape(Y, X)
    out = gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, 25)
    _check(out, ref)
    for w in ("0", "3", "6"):     # forced tile shapes give the same values
        os.environ["GPP_TILE_WSHIFT"] = w
        try:
            out_w = gridpp.optimal_interpolation(gridpp.Grid(lats, lons), bg, points, obs, ratios, pbg, st, 25)
        finally:
            del os.environ["GPP_TILE_WSHIFT"]
        _check(out_w, ref)


@pytest.mark.parametrize("seed", range(600, 616))
def test_random_configurations_spatially_varying_barnes_on_the_tile_path(seed, monkeypatch):
    """k_oi_union_sp (round 6): a Barnes structure whose scales vary in space -- P is not symmetric (corr(p1, p2) takes the scales at p1,
    structure.cpp:188-214), the shared factor of a tile is an unpivoted LU of its core block.  Random grids and densities (partial tiles, unions
    from a handful to far beyond the 40-row limit: the 16-cell / 4-cell list passes and the pivoted LU of k_oi behind them), smooth and
    white-noise scale fields of +-30 %, with and without vertical / laf scales, missing elevations, anti-extrapolation, max_points 1..32 --
    against the oracle, and bit for bit nothing: the pivoted LU per selection (GPP_OI_NO_SP_UNION) is a different elimination order."""
    import gridpp_amd as gridpp
    from oracle import oracle as O
    rng = np.random.default_rng(9000 + seed)
    Y, X = int(rng.integers(9, 70)), int(rng.integers(9, 70))
    S = int(rng.choice([12, 60, 250, 900]))
    base = float(rng.choice([4000.0, 9000.0, 20000.0]))
    mp = int(rng.choice([1, 3, 10, 20, 30, 32]))
    ext = float(rng.choice([0.2, 0.5, 1.0]))
    lats, lons = np.meshgrid(np.linspace(60, 60 + ext, Y), np.linspace(10, 10 + 2 * ext, X), indexing="ij")
    with_v = seed % 3 != 0
    ge = (200 + 150 * np.sin(7 * lats) * np.cos(5 * lons)).astype(np.float32) if with_v else np.zeros((Y, X), np.float32)
    gl = np.clip(0.5 + 0.5 * np.sin(11 * lons), 0, 1).astype(np.float32) if with_v else np.zeros((Y, X), np.float32)
    plat, plon = 60 + ext * rng.random(S), 10 + 2 * ext * rng.random(S)
    pe = rng.uniform(50, 350, S).astype(np.float32) if with_v else np.zeros(S, np.float32)
    pl = rng.uniform(0, 1, S).astype(np.float32) if with_v else np.zeros(S, np.float32)
    if with_v and seed % 4 == 1:
        pe[rng.random(S) < 0.1] = np.nan
    if seed % 2:
        hf = (base * rng.uniform(0.7, 1.3, (Y, X))).astype(np.float32)
    else:
        hf = (base * (1 + 0.3 * np.sin(9 * lats) * np.cos(6 * lons))).astype(np.float32)
    vf = (300 * rng.uniform(0.8, 1.2, (Y, X))).astype(np.float32) if with_v else np.zeros((Y, X), np.float32)
    wf = (0.6 * rng.uniform(0.8, 1.2, (Y, X))).astype(np.float32) if (with_v and seed % 2) else np.zeros((Y, X), np.float32)
    bg = rng.normal(0, 2, (Y, X)).astype(np.float32)
    if seed % 5 == 2:
        bg[rng.random((Y, X)) < 0.02] = np.nan
    obs, pbg = rng.normal(0, 2, S).astype(np.float32), rng.normal(0, 2, S).astype(np.float32)
    ratios = rng.uniform(0.05, 2, S).astype(np.float32)
    allow = bool(seed % 2)
    min_rho = 0.0013
    grid = gridpp.Grid(lats, lons, ge, gl)
    points = gridpp.Points(plat, plon, pe, pl)
    st = gridpp.BarnesStructure(grid, hf, vf, wf, min_rho)
    out = np.asarray(gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, mp, allow))
    stats = gridpp.oi_last_stats()
    og = O.Pts(lats.ravel(), lons.ravel(), ge.ravel(), gl.ravel())
    op = O.Pts(plat, plon, pe, pl)
    ci, oi = np.arange(Y * X), O.nearest_indices(og, op)
    Rf = np.array([O.structure_localization("Barnes", h, min_rho) for h in hf.ravel()], np.float32)
    cp = [a.ravel()[ci] for a in (hf, vf, wf)] + [Rf[ci]]
    opar = [a.ravel()[oi] for a in (hf, vf, wf)] + [Rf[oi]]
    ones_g, ones_p = np.ones(Y * X, np.float32), np.ones(S, np.float32)
    ref, _ = O.oi_full_generic(og, bg.ravel(), ones_g, op, obs, ratios, pbg, ones_p, O.Struct("Barnes", base), mp, allow, cp, opar)
    _check(out, ref.reshape(Y, X))
    assert stats["union_kernel_ms"] > 0          # (the tile path ran)
    monkeypatch.setenv("GPP_OI_NO_SP_UNION", "1")
    out2 = np.asarray(gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, mp, allow))
    _check(out2, ref.reshape(Y, X))
