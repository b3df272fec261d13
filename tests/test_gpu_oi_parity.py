"""HIP optimal_interpolation vs the CPU oracle on seeded inputs (through the C-ABI).

Tolerance (BASELINE.json north_star): 1e-5 relative for OI floats.  Cells whose top-max_points
cut straddles an exact float-rho tie are implementation-defined in the reference
(src/api/oi.cpp:266, unstable std::sort); the oracle and the kernel share one tie-break
(lower observation index), so they are compared too."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def rel_err(a, b, floor=1e-3):
    return np.max(np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b), floor))


def make_case(seed, Y, X, S, geodetic=True, with_elev=False, nan_frac=0.0, dup=False):
    rng = np.random.default_rng(seed)
    if geodetic:
        lats, lons = np.meshgrid(np.linspace(0, 1, Y), np.linspace(0, 1, X), indexing="ij")
        plat, plon = rng.random(S), rng.random(S)
    else:
        lats, lons = np.meshgrid(np.linspace(0, 100000, Y), np.linspace(0, 100000, X), indexing="ij")
        plat, plon = rng.random(S) * 100000, rng.random(S) * 100000
    if dup and S > 4:
        plat[1], plon[1] = plat[0], plon[0]
    bg = rng.normal(0, 1, (Y, X)).astype(np.float32)
    obs = rng.normal(0, 1, S).astype(np.float32)
    pbg = rng.normal(0, 1, S).astype(np.float32)
    ratios = rng.uniform(0.1, 1, S).astype(np.float32)
    gelev = glaf = pelev = plaf = None
    if with_elev:
        gelev, glaf = rng.uniform(0, 1000, (Y, X)), rng.uniform(0, 1, (Y, X))
        pelev, plaf = rng.uniform(0, 1000, S), rng.uniform(0, 1, S)
    if nan_frac > 0:
        obs[rng.random(S) < nan_frac] = np.nan
        pbg[rng.random(S) < nan_frac] = np.nan
        bg[rng.random((Y, X)) < nan_frac] = np.nan
    return dict(lats=lats, lons=lons, plat=plat, plon=plon, bg=bg, obs=obs, pbg=pbg, ratios=ratios,
                gelev=gelev, glaf=glaf, pelev=pelev, plaf=plaf, ctype=0 if geodetic else 1)


def run_both(c, h, v, w, max_points, allow_extrap=True, full=False):
    import gridpp_amd as gridpp
    from oracle import oracle as O
    Y, X = c["bg"].shape
    e = (c["gelev"], c["glaf"]) if c["gelev"] is not None else ((), ())
    pe = (c["pelev"], c["plaf"]) if c["pelev"] is not None else ((), ())
    grid = gridpp.Grid(c["lats"], c["lons"], e[0], e[1], c["ctype"])
    points = gridpp.Points(c["plat"], c["plon"], pe[0], pe[1], c["ctype"])
    st = gridpp.BarnesStructure(h, v, w)
    og = O.Pts(c["lats"].ravel(), c["lons"].ravel(), None if c["gelev"] is None else c["gelev"].ravel(),
               None if c["glaf"] is None else c["glaf"].ravel(), c["ctype"])
    op = O.Pts(c["plat"], c["plon"], c["pelev"], c["plaf"], c["ctype"])
    ost = O.Barnes(h, v, w)
    if full:
        rng = np.random.default_rng(7)
        bvar = rng.uniform(0.5, 2, (Y, X)).astype(np.float32)
        bvp = rng.uniform(0.5, 2, c["obs"].size).astype(np.float32)
        out, var = gridpp.optimal_interpolation_full(grid, c["bg"], bvar, points, c["obs"], c["ratios"], c["pbg"], bvp, st,
                                                     max_points, allow_extrap)
        ref, rvar = O.oi_full(og, c["bg"].ravel(), bvar.ravel(), op, c["obs"], c["ratios"], c["pbg"], bvp, ost, max_points,
                              allow_extrap)
        return out, ref.reshape(Y, X), var, rvar.reshape(Y, X)
    out = gridpp.optimal_interpolation(grid, c["bg"], points, c["obs"], c["ratios"], c["pbg"], st, max_points, allow_extrap)
    ref = O.oi(og, c["bg"].ravel(), op, c["obs"], c["ratios"], c["pbg"], ost, max_points, allow_extrap).reshape(Y, X)
    return out, ref


def check(out, ref):
    assert out.dtype == np.float32 and out.shape == ref.shape
    assert (np.isnan(out) == np.isnan(ref)).all()
    m = ~np.isnan(ref)
    assert rel_err(out[m], ref[m]) < RTOL


@pytest.mark.parametrize("max_points", [1, 5, 20, 30])
def test_geodetic_benchmark_style(max_points):
    c = make_case(1001, 64, 64, 200)
    out, ref = run_both(c, 10000, 0, 0, max_points)
    check(out, ref)
    assert np.abs(out - c["bg"]).max() > 0.1   # the analysis actually moved


def test_readme_config1():
    # BASELINE.json configs[0]: 200x200 grid, 10 obs, BarnesStructure(10000), max_points=10
    c = make_case(1000, 200, 200, 10)
    c["ratios"][:] = 0.5
    out, ref = run_both(c, 10000, 0, 0, 10)
    check(out, ref)


def test_odd_shapes_and_points_overload():
    import gridpp_amd as gridpp
    from oracle import oracle as O
    c = make_case(5, 37, 53, 150)
    out, ref = run_both(c, 12000, 0, 0, 12)
    check(out, ref)
    # Points (1-D) overload on a shuffled point list: tiles have no spatial coherence
    rng = np.random.default_rng(3)
    n = 1000
    blat, blon = rng.random(n), rng.random(n)
    bgp = rng.normal(0, 1, n).astype(np.float32)
    bpoints = gridpp.Points(blat, blon)
    points = gridpp.Points(c["plat"], c["plon"])
    out = gridpp.optimal_interpolation(bpoints, bgp, points, c["obs"], c["ratios"], c["pbg"], gridpp.BarnesStructure(12000), 12)
    ref = O.oi(O.Pts(blat, blon), bgp, O.Pts(c["plat"], c["plon"]), c["obs"], c["ratios"], c["pbg"], O.Barnes(12000), 12)
    check(out, ref)


def test_elev_laf_structure():
    c = make_case(11, 48, 48, 300, with_elev=True)
    out, ref = run_both(c, 10000, 200, 0.5, 15)
    check(out, ref)


def test_cartesian_nan_duplicates_no_extrapolation():
    c = make_case(21, 40, 56, 250, geodetic=False, nan_frac=0.05, dup=True)
    out, ref = run_both(c, 9000, 0, 0, 10, allow_extrap=False)
    check(out, ref)


def test_full_variance():
    c = make_case(31, 48, 40, 200, nan_frac=0.02)
    out, ref, var, rvar = run_both(c, 10000, 0, 0, 16, full=True)
    check(out, ref)
    check(var, rvar)


def test_few_obs_keep_all():
    # fewer usable observations than max_points: all are kept (src/api/oi.cpp:274-281)
    c = make_case(41, 32, 32, 8)
    out, ref = run_both(c, 30000, 0, 0, 30)
    check(out, ref)


def test_selection_is_bit_exact_vs_oracle_neighbours():
    """Radius membership on the device path equals the oracle's index set (bit-exact integers)."""
    import gridpp_amd as gridpp
    from oracle import oracle as O
    c = make_case(51, 8, 8, 500)
    points = gridpp.Points(c["plat"], c["plon"])
    op = O.Pts(c["plat"], c["plon"])
    for lat, lon, r in [(0.5, 0.5, 20000.0), (0.1, 0.9, 36456.5), (0.0, 0.0, 5000.0)]:
        np.testing.assert_array_equal(points.get_neighbours(lat, lon, r), O.get_neighbours(op, lat, lon, r))


# ---- structure functions other than Barnes (scalar forms) ---------------------------------------
def _generic(c, make_gpu, make_orc, max_points, full=False):
    import gridpp_amd as gridpp
    from oracle import oracle as O
    Y, X = c["bg"].shape
    e = (c["gelev"], c["glaf"]) if c["gelev"] is not None else ((), ())
    pe = (c["pelev"], c["plaf"]) if c["pelev"] is not None else ((), ())
    grid = gridpp.Grid(c["lats"], c["lons"], e[0], e[1], c["ctype"])
    points = gridpp.Points(c["plat"], c["plon"], pe[0], pe[1], c["ctype"])
    og = O.Pts(c["lats"].ravel(), c["lons"].ravel(), None if c["gelev"] is None else c["gelev"].ravel(),
               None if c["glaf"] is None else c["glaf"].ravel(), c["ctype"])
    op = O.Pts(c["plat"], c["plon"], c["pelev"], c["plaf"], c["ctype"])
    ones_g, ones_p = np.ones((Y, X), np.float32), np.ones(c["obs"].size, np.float32)
    out, var = gridpp.optimal_interpolation_full(grid, c["bg"], ones_g, points, c["obs"], c["ratios"], c["pbg"], ones_p,
                                                 make_gpu(gridpp), max_points)
    ref, rvar = O.oi_full_generic(og, c["bg"].ravel(), ones_g.ravel(), op, c["obs"], c["ratios"], c["pbg"], ones_p, make_orc(O), max_points)
    # the same call without a variance output (on the pivoted-LU path: one solve per selection and a dot product per cell)
    plain = gridpp.optimal_interpolation(grid, c["bg"], points, c["obs"], c["ratios"], c["pbg"], make_gpu(gridpp), max_points)
    check(np.asarray(plain), ref.reshape(Y, X))
    return np.asarray(out), ref.reshape(Y, X), np.asarray(var), rvar.reshape(Y, X)


@pytest.mark.parametrize("kind", ["Cressman", "Soar", "Toar", "Powerlaw"])
@pytest.mark.parametrize("elev", [False, True])
def test_other_kernels(kind, elev):
    """Horizontal-only (symmetric, Cholesky) and with vertical / laf factors (Cressman/SOAR/TOAR on signed differences
    are not even functions -> non-symmetric P -> pivoted LU path, like the reference's arma::inv)."""
    c = make_case(61, 40, 36, 120, with_elev=elev)
    h = {"Cressman": 30000, "Soar": 4000, "Toar": 3500, "Powerlaw": 2500}[kind]
    v, w = (300, 0.6) if elev else (0, 0)
    out, ref, var, rvar = _generic(c, lambda g: getattr(g, kind + "Structure")(h, v, w), lambda O: O.Struct(kind, h, v, w), 12)
    check(out, ref)
    check(var, rvar)
    assert np.abs(out - c["bg"]).max() > 0.05


def test_multiple_and_cross_validation_structures():
    c = make_case(62, 36, 40, 150, with_elev=True)
    mk_g = lambda g: g.MultipleStructure(g.BarnesStructure(12000), g.LinearStructure(0, 0.2, 0), g.PowerlawStructure(1, 1, 0.7))
    mk_o = lambda O: O.Struct.multiple(O.Struct("Barnes", 12000), O.Struct("Linear", 0, 0.2, 0), O.Struct("Powerlaw", 1, 1, 0.7))
    out, ref, var, rvar = _generic(c, mk_g, mk_o, 14)
    check(out, ref)
    check(var, rvar)
    out, ref, var, rvar = _generic(c, lambda g: g.CrossValidation(g.BarnesStructure(12000, 200, 0.5), 3000),
                                   lambda O: O.Struct("Barnes", 12000, 200, 0.5).cross_validation(3000), 14)
    check(out, ref)
    # a MultipleStructure as a component of a MultipleStructure (the reference delegates, structure.cpp:90-138): the vertical
    # factor is the inner structure's vertical component, the laf factor the inner structure's laf component
    mk_g = lambda g: g.MultipleStructure(g.BarnesStructure(12000),
                                         g.MultipleStructure(g.BarnesStructure(5000), g.LinearStructure(0, 0.2, 0), g.BarnesStructure(1, 1, 9)),
                                         g.MultipleStructure(g.CressmanStructure(7000), g.BarnesStructure(1, 77, 1), g.PowerlawStructure(1, 1, 0.7)))
    mk_o = lambda O: O.Struct.multiple(O.Struct("Barnes", 12000),
                                       O.Struct.multiple(O.Struct("Barnes", 5000), O.Struct("Linear", 0, 0.2, 0), O.Struct("Barnes", 1, 1, 9)),
                                       O.Struct.multiple(O.Struct("Cressman", 7000), O.Struct("Barnes", 1, 77, 1), O.Struct("Powerlaw", 1, 1, 0.7)))
    out, ref, var, rvar = _generic(c, mk_g, mk_o, 14)
    check(out, ref)
    check(var, rvar)


def test_pivoted_lu_equals_cholesky(monkeypatch):
    """The LU solver variant (used for non-symmetric / indefinite systems) gives the Cholesky answer on an SPD system."""
    c = make_case(63, 32, 32, 100)
    out, ref = run_both(c, 10000, 0, 0, 16)
    monkeypatch.setenv("GPP_OI_FORCE_LU", "1")
    out_lu, _ = run_both(c, 10000, 0, 0, 16)
    check(out_lu, ref)
    assert np.max(np.abs(out_lu - out)) < 1e-6


@pytest.mark.parametrize("max_points", [33, 48, 62, 0, 100])
def test_more_than_32_points(max_points):
    """max_points in 33..62 (and 0 / larger values while no cell has more than 62 usable observations) use the
    62-row register tile."""
    S = 150 if max_points in (33, 48, 62) else 55
    c = make_case(70 + max_points, 24, 24, S)
    out, ref = run_both(c, 40000, 0, 0, max_points)   # R = 146 km: every observation is in range of every cell
    check(out, ref)


@pytest.mark.parametrize("max_points,S", [(0, 200), (100, 300), (0, 450), (63, 150)])
def test_more_than_62_points_large_n_kernel(max_points, S):
    """More usable observations per grid point than the 62-row register tile holds (max_points == 0 with every observation
    in range, or max_points > 62): k_oi_big, one workgroup per grid point."""
    import gridpp_amd as gridpp
    c = make_case(80 + S, 9, 11, S)
    out, ref = run_both(c, 40000, 0, 0, max_points)   # R = 146 km: every observation is in range of every cell
    check(out, ref)
    assert gridpp.oi_last_stats()["big_cells"] == 99
    out, ref, var, rvar = run_both(c, 40000, 0, 0, max_points, allow_extrap=False, full=True)
    check(out, ref)
    check(var, rvar)


def test_more_points_than_the_large_n_kernel_holds():
    """700 usable observations at every grid point (max_points = 0): beyond k_oi_big's 512, taken by the general kernel
    (k_oi_huge: pivoted elimination in HBM scratch) -- the reference has no limit (oi.cpp:262-315)."""
    c = make_case(80, 4, 4, 700)
    out, ref, var, rvar = run_both(c, 40000, 0, 0, 0, full=True)
    check(out, ref)
    check(var, rvar)
    out, ref = run_both(c, 40000, 0, 0, 600, allow_extrap=False)
    check(out, ref)


def test_large_n_with_the_pivoted_solver(monkeypatch):
    """More than 62 usable observations on the LU path (non-symmetric systems, retry after a non-positive pivot): the general kernel."""
    import gridpp_amd as gridpp
    c = make_case(83, 7, 9, 180)
    monkeypatch.setenv("GPP_OI_FORCE_LU", "1")
    out, ref, var, rvar = run_both(c, 40000, 0, 0, 0, full=True)
    check(out, ref)
    check(var, rvar)
    assert gridpp.oi_last_stats()["big_cells"] == 63
    out, ref = run_both(c, 40000, 0, 0, 100, allow_extrap=False)
    check(out, ref)


@pytest.mark.parametrize("kind", ["Barnes", "Soar"])
def test_spatially_varying_structure_large_n(kind):
    """A spatially varying structure with more than 62 usable observations per grid point (max_points = 0 and 80)."""
    import gridpp_amd as gridpp
    from oracle import oracle as O
    c = make_case(92, 10, 12, 140, with_elev=True)
    Y, X = c["bg"].shape
    rng = np.random.default_rng(6)
    base = {"Barnes": 30000, "Soar": 9000}[kind]
    hf = (base * rng.uniform(0.8, 1.2, (Y, X))).astype(np.float32)
    vf = (300 * rng.uniform(0.7, 1.3, (Y, X))).astype(np.float32)
    wf = (0.6 * rng.uniform(0.7, 1.3, (Y, X))).astype(np.float32)
    min_rho = 0.0013
    grid = gridpp.Grid(c["lats"], c["lons"], c["gelev"], c["glaf"])
    points = gridpp.Points(c["plat"], c["plon"], c["pelev"], c["plaf"])
    st = getattr(gridpp, kind + "Structure")(grid, hf, vf, wf, min_rho)
    ones_g, ones_p = np.ones((Y, X), np.float32), np.ones(c["obs"].size, np.float32)
    og = O.Pts(c["lats"].ravel(), c["lons"].ravel(), c["gelev"].ravel(), c["glaf"].ravel())
    op = O.Pts(c["plat"], c["plon"], c["pelev"], c["plaf"])
    ci, oi = np.arange(Y * X), O.nearest_indices(og, op)
    Rf = np.array([O.structure_localization(kind, h, min_rho) for h in hf.ravel()], np.float32)
    cp = [a.ravel()[ci] for a in (hf, vf, wf)] + [Rf[ci]]
    opar = [a.ravel()[oi] for a in (hf, vf, wf)] + [Rf[oi]]
    ost = O.Struct(kind, base)
    for mp in (0, 80):
        out, var = gridpp.optimal_interpolation_full(grid, c["bg"], ones_g, points, c["obs"], c["ratios"], c["pbg"], ones_p, st, mp)
        ref, rvar = O.oi_full_generic(og, c["bg"].ravel(), ones_g.ravel(), op, c["obs"], c["ratios"], c["pbg"], ones_p, ost, mp, True, cp, opar)
        check(np.asarray(out), ref.reshape(Y, X))
        check(np.asarray(var), rvar.reshape(Y, X))
        assert gridpp.oi_last_stats()["big_cells"] > 0


# ---- spatially varying structure functions (structure.cpp:168-214) ------------------------------------------------------
def test_spatial_barnes_pin():
    """tests/test_barnes_structure.py:46-66"""
    import gridpp_amd as gridpp
    y, x = [[0, 0]], [[0, 2500]]
    grid = gridpp.Grid(y, x, y, y, gridpp.Cartesian)
    min_rho = 0.1
    st = gridpp.BarnesStructure(grid, [[2500, 1]], [[0, 0]], [[0, 0]], min_rho)
    p1 = gridpp.Point(0, 0, 0, 0, gridpp.Cartesian)
    p2 = gridpp.Point(0, 2500, 0, 0, gridpp.Cartesian)
    assert abs(st.localization_distance(p1) - np.sqrt(-2 * np.log(min_rho)) * 2500) < 5e-5 * 2500
    assert abs(st.corr(p1, p2) - 0.6) < 0.05
    assert abs(st.localization_distance(p2) - np.sqrt(-2 * np.log(min_rho)) * 1) < 5e-5
    assert abs(st.corr(p2, p1) - 0) < 0.05
    yy, xx = np.meshgrid(np.linspace(0, 1, 2), np.linspace(0, 1, 3))
    g2 = gridpp.Grid(yy, xx, yy, yy, gridpp.Cartesian)
    valid = np.ones([3, 2])
    gridpp.BarnesStructure(g2, valid, valid, valid)
    for inval in (np.ones([3, 4]), np.ones([2, 2]), np.ones([2, 4])):
        for args in ((inval, valid, valid), (valid, inval, valid), (valid, valid, inval)):
            with pytest.raises(ValueError):
                gridpp.BarnesStructure(g2, *args)


@pytest.mark.parametrize("kind", ["Barnes", "Soar", "Powerlaw"])
@pytest.mark.parametrize("same_grid", [True, False])
def test_spatially_varying_structure(kind, same_grid):
    import gridpp_amd as gridpp
    from oracle import oracle as O
    c = make_case(90, 30, 34, 100, with_elev=True)
    Y, X = c["bg"].shape
    rng = np.random.default_rng(5)
    if same_grid:
        flat_, flon_ = c["lats"], c["lons"]
    else:   # a coarser field grid: nearest-neighbour lookup for every grid point and observation
        flat_, flon_ = np.meshgrid(np.linspace(0, 1, 9), np.linspace(0, 1, 11), indexing="ij")
    base = {"Barnes": 9000, "Soar": 2500, "Powerlaw": 2000}[kind]
    hf = (base * rng.uniform(0.7, 1.3, flat_.shape)).astype(np.float32)
    vf = (300 * rng.uniform(0.7, 1.3, flat_.shape)).astype(np.float32)
    wf = (0.6 * rng.uniform(0.7, 1.3, flat_.shape)).astype(np.float32)
    min_rho = 0.0013
    grid = gridpp.Grid(c["lats"], c["lons"], c["gelev"], c["glaf"])
    fgrid = grid if same_grid else gridpp.Grid(flat_, flon_)
    points = gridpp.Points(c["plat"], c["plon"], c["pelev"], c["plaf"])
    st = getattr(gridpp, kind + "Structure")(fgrid, hf, vf, wf, min_rho)
    ones_g, ones_p = np.ones((Y, X), np.float32), np.ones(c["obs"].size, np.float32)
    out, var = gridpp.optimal_interpolation_full(grid, c["bg"], ones_g, points, c["obs"], c["ratios"], c["pbg"], ones_p, st, 10)
    # oracle: parameters at the nearest field point of every background point / observation
    og = O.Pts(c["lats"].ravel(), c["lons"].ravel(), c["gelev"].ravel(), c["glaf"].ravel())
    op = O.Pts(c["plat"], c["plon"], c["pelev"], c["plaf"])
    of = O.Pts(flat_.ravel(), flon_.ravel())
    ci, oi = O.nearest_indices(of, og), O.nearest_indices(of, op)
    Rf = np.array([O.structure_localization(kind, h, min_rho) for h in hf.ravel()], np.float32)
    cp = [a.ravel()[ci] for a in (hf, vf, wf)] + [Rf[ci]]
    opar = [a.ravel()[oi] for a in (hf, vf, wf)] + [Rf[oi]]
    ost = O.Struct(kind, base)
    ref, rvar = O.oi_full_generic(og, c["bg"].ravel(), ones_g.ravel(), op, c["obs"], c["ratios"], c["pbg"], ones_p, ost, 10,
                                  True, cp, opar)
    check(np.asarray(out), ref.reshape(Y, X))
    check(np.asarray(var), rvar.reshape(Y, X))
    assert np.abs(np.asarray(out) - c["bg"]).max() > 0.05
    # without a variance output the pivoted-LU path solves (P+R) z = d once per selection and leaves every cell the dot product G.z
    out2 = gridpp.optimal_interpolation(grid, c["bg"], points, c["obs"], c["ratios"], c["pbg"], st, 10)
    check(np.asarray(out2), ref.reshape(Y, X))
    out3 = gridpp.optimal_interpolation(grid, c["bg"], points, c["obs"], c["ratios"], c["pbg"], st, 10, False)
    ref3, _ = O.oi_full_generic(og, c["bg"].ravel(), ones_g.ravel(), op, c["obs"], c["ratios"], c["pbg"], ones_p, ost, 10,
                                False, cp, opar)
    check(np.asarray(out3), ref3.reshape(Y, X))


@pytest.mark.parametrize("allow", [True, False])
def test_62_row_tile_groups_share_one_factorisation(allow):
    """max_points in 33..62 with every grid point selecting the same observations: one factorisation per tile, the other
    63 cells reuse it through forward substitutions (k_oi<62>)."""
    import gridpp_amd as gridpp
    c = make_case(91, 40, 40, 45)
    out, ref, var, rvar = run_both(c, 40000, 0, 0, 60, allow_extrap=allow, full=True)   # all 45 observations are in range everywhere
    check(out, ref)
    check(var, rvar)
    s = gridpp.oi_last_stats()
    assert s["solves"] <= 2 * 25          # 25 tiles, one (rarely two) selections each


def test_spatially_varying_parts_inside_a_multiple_structure():
    """MultipleStructure(sh, sv, sw) with spatially varying parts on three DIFFERENT grids (structure.cpp:90-138): per point the
    horizontal scale and the localization distance are sh's, the vertical scale sv's, the laf scale sw's."""
    import gridpp_amd as gridpp
    from oracle import oracle as O
    c = make_case(95, 26, 30, 90, with_elev=True)
    Y, X = c["bg"].shape
    rng = np.random.default_rng(8)

    def field_grid(ny, nx):
        la, lo = np.meshgrid(np.linspace(0, 1, ny), np.linspace(0, 1, nx), indexing="ij")
        return la, lo
    (hla, hlo), (vla, vlo), (wla, wlo) = field_grid(7, 9), field_grid(5, 6), field_grid(11, 4)
    hf = (9000 * rng.uniform(0.7, 1.3, hla.shape)).astype(np.float32)
    vf = (300 * rng.uniform(0.7, 1.3, vla.shape)).astype(np.float32)
    wf = (0.6 * rng.uniform(0.7, 1.3, wla.shape)).astype(np.float32)
    one = lambda a: np.ones(a.shape, np.float32)
    gh, gv, gw = gridpp.Grid(hla, hlo), gridpp.Grid(vla, vlo), gridpp.Grid(wla, wlo)
    sh = gridpp.BarnesStructure(gh, hf, 100 * one(hf), 0.3 * one(hf), 0.0013)
    sv = gridpp.SoarStructure(gv, 5000 * one(vf), vf, 0.3 * one(vf), 0.0013)
    sw = gridpp.BarnesStructure(gw, 5000 * one(wf), 100 * one(wf), wf, 0.0013)
    st = gridpp.MultipleStructure(sh, sv, sw)
    grid = gridpp.Grid(c["lats"], c["lons"], c["gelev"], c["glaf"])
    points = gridpp.Points(c["plat"], c["plon"], c["pelev"], c["plaf"])
    ones_g, ones_p = np.ones((Y, X), np.float32), np.ones(c["obs"].size, np.float32)
    og = O.Pts(c["lats"].ravel(), c["lons"].ravel(), c["gelev"].ravel(), c["glaf"].ravel())
    op = O.Pts(c["plat"], c["plon"], c["pelev"], c["plaf"])
    par = []
    for pts in (og, op):
        ih = O.nearest_indices(O.Pts(hla.ravel(), hlo.ravel()), pts)
        iv = O.nearest_indices(O.Pts(vla.ravel(), vlo.ravel()), pts)
        iw = O.nearest_indices(O.Pts(wla.ravel(), wlo.ravel()), pts)
        Rf = np.array([O.structure_localization("Barnes", h, 0.0013) for h in hf.ravel()], np.float32)
        par.append([hf.ravel()[ih], vf.ravel()[iv], wf.ravel()[iw], Rf[ih]])
    ost = O.Struct.multiple(O.Struct("Barnes", 9000), O.Struct("Soar", 5000, 300), O.Struct("Barnes", 5000, 0, 0.6))
    for mp in (10, 0):
        out, var = gridpp.optimal_interpolation_full(grid, c["bg"], ones_g, points, c["obs"], c["ratios"], c["pbg"], ones_p, st, mp)
        ref, rvar = O.oi_full_generic(og, c["bg"].ravel(), ones_g.ravel(), op, c["obs"], c["ratios"], c["pbg"], ones_p, ost, mp, True, par[0], par[1])
        check(np.asarray(out), ref.reshape(Y, X))
        check(np.asarray(var), rvar.reshape(Y, X))
    assert np.abs(np.asarray(out) - c["bg"]).max() > 0.05
    # the single-pair functions see the same scales
    p1 = gridpp.Point(0.31, 0.42, 120.0, 0.4)
    p2 = gridpp.Point(0.33, 0.45, 260.0, 0.7)
    q1, q2 = O.Pts([0.31], [0.42], [120.0], [0.4]), O.Pts([0.33], [0.45], [260.0], [0.7])
    k = [int(O.nearest_indices(O.Pts(a.ravel(), b.ravel()), q1)[0]) for a, b in ((hla, hlo), (vla, vlo), (wla, wlo))]
    s1 = O.Struct.multiple(O.Struct("Barnes", float(hf.ravel()[k[0]])), O.Struct("Soar", 5000, float(vf.ravel()[k[1]])), O.Struct("Barnes", 5000, 0, float(wf.ravel()[k[2]])))
    want = s1.corr((q1.x[0], q1.y[0], q1.z[0], 120.0, 0.4), (q2.x[0], q2.y[0], q2.z[0], 260.0, 0.7))
    assert abs(st.corr(p1, p2) - want) < 1e-6


# ---- LAPACK golden vectors (tools/make_oi_fixtures.py: independent numpy + scipy restatement of src/api/oi.cpp:176-338) -----------
from tests import oi_golden  # noqa: E402


@pytest.mark.parametrize("name", oi_golden.NAMES)
def test_oi_golden_vectors(name):
    import gridpp_amd as gridpp
    c = oi_golden.CASES[name]
    h, v, w, mp, allow = c["params"]
    points = gridpp.Points(c["plat"], c["plon"], c["pelev"], c["plaf"])
    st = gridpp.CressmanStructure(h, v, w) if int(c["kind"]) == 1 else gridpp.BarnesStructure(h, v, w)
    if c["shape"][0] > 0:
        Y, X = int(c["shape"][0]), int(c["shape"][1])
        grid = gridpp.Grid(c["blat"].reshape(Y, X), c["blon"].reshape(Y, X), c["belev"].reshape(Y, X), c["blaf"].reshape(Y, X))
        out, var = gridpp.optimal_interpolation_full(grid, c["background"].reshape(Y, X), c["bvariance"].reshape(Y, X), points, c["pobs"],
                                                     c["obs_variance"], c["pbackground"], c["bvariance_at_points"], st, int(mp), bool(allow))
    else:
        bpoints = gridpp.Points(c["blat"], c["blon"], c["belev"], c["blaf"])
        out, var = gridpp.optimal_interpolation_full(bpoints, c["background"], c["bvariance"], points, c["pobs"], c["obs_variance"],
                                                     c["pbackground"], c["bvariance_at_points"], st, int(mp), bool(allow))
    oi_golden.check(out, var, c)
