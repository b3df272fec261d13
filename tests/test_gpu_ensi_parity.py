"""HIP optimal_interpolation_ensi vs the CPU oracle (oracle/gridpp_oracle.c restates src/api/oi_ensi.cpp with a
partial-pivot inverse + Jacobi eig in double) and vs the LAPACK golden vectors of tests/golden/ensi_cases.npz.  The reference's
own tests hold no numeric EnSI value (tests/test_optimal_interpolation_ens.py:9-35: two pass-through cases); the oracle is pinned
by the independent numpy + scipy/LAPACK restatement that wrote the golden vectors (tools/make_ensi_fixtures.py).
Tolerance 1e-5 relative: the plain reading |out - ref| / max(|ref|, 1e-2) is asserted with the sweeps run to convergence
(test_strict_measure_*), the default fast path is held to it up to one float32 ulp of a cell's members (ensi_golden.rel_err)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def case(seed, Y, X, E, S, nan_member=None, nan_obs=False):
    rng = np.random.default_rng(seed)
    lats, lons = np.meshgrid(np.linspace(0, 1, Y), np.linspace(0, 1, X), indexing="ij")
    base = np.sin(5 * lats) * np.cos(3 * lons)
    bg = (base[:, :, None] + rng.normal(0, 1, (Y, X, E))).astype(np.float32)
    plat, plon = rng.random(S), rng.random(S)
    pbg = rng.normal(0, 1, (S, E)).astype(np.float32)
    obs = rng.normal(0, 1, S).astype(np.float32)
    sig = rng.uniform(0.5, 2, S).astype(np.float32)
    if nan_member is not None:
        bg[3, 4, nan_member] = np.nan
    if nan_obs:
        obs[::7] = np.nan
    return lats, lons, bg, plat, plon, pbg, obs, sig


def run(c, h, max_points, allow=True, v=0, w=0, elev=False, want_ref=True):
    import gridpp_amd as gridpp
    from oracle import oracle as O
    lats, lons, bg, plat, plon, pbg, obs, sig = c
    Y, X, E = bg.shape
    rng = np.random.default_rng(99)
    ge = rng.uniform(0, 500, (Y, X)) if elev else ()
    pe = rng.uniform(0, 500, plat.size) if elev else ()
    grid = gridpp.Grid(lats, lons, ge, ())
    points = gridpp.Points(plat, plon, pe, ())
    out = gridpp.optimal_interpolation_ensi(grid, bg, points, obs, sig, pbg, gridpp.BarnesStructure(h, v, w), max_points, allow)
    if not want_ref:
        return np.asarray(out), None
    og = O.Pts(lats.ravel(), lons.ravel(), ge.ravel() if elev else None)
    op = O.Pts(plat, plon, pe if elev else None)
    ref = O.oi_ensi(og, bg.reshape(-1, E), op, obs, sig, pbg, O.Barnes(h, v, w), max_points, allow).reshape(Y, X, E)
    return np.asarray(out), ref


def check(out, ref, bg):
    assert out.dtype == np.float32 and out.shape == ref.shape
    assert (np.isnan(out) == np.isnan(ref)).all()
    from tests.ensi_golden import rel_err
    err = rel_err(out, ref, bg)     # relative to max(|ref|, 1e-2, one float32 ulp of the cell's members)
    assert err.max() < RTOL, err.max()
    assert np.nanmax(np.abs(out - bg)) > 0.05   # the update did something


def plain_err(out, ref):
    """north_star's reading: |out - ref| / max(|ref|, 1e-2)"""
    m = ~np.isnan(ref)
    return (np.abs(out.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-2))[m]


def test_strict_measure_with_converged_sweeps_and_count_for_the_fast_path():
    """Config-5-like inputs (E = 50, max_points 30, h = 10 km on a 1 x 1 degree domain, dense observations).
    (a) gpp_ensi_set_convergence(1): the Jacobi sweeps run to convergence and the PLAIN 1e-5 measure holds for every value.
    (b) default (sweeps stopped at 0.040 c + the float32 perturbation series of round 4 (six products since round 6)): the PLAIN measure holds as well (round 3,
        with three products and 0.010 c, left about one value in 10^6 outside it by one float32 ulp of a term: DESIGN.md 4.2)."""
    import gridpp_amd as gridpp
    c = case(4242, 56, 60, 50, 1200)
    try:
        gridpp.ensi_set_convergence(True)
        out, ref = run(c, 10000, 30)
    finally:
        gridpp.ensi_set_convergence(False)
    assert (np.isnan(out) == np.isnan(ref)).all()
    e = plain_err(out, ref)
    assert e.max() < RTOL, e.max()
    assert np.nanmax(np.abs(out - c[2])) > 0.05
    out2, _ = run(c, 10000, 30)
    e2 = plain_err(out2, ref)
    assert e2.max() < RTOL, (e2.max(), int((e2 >= RTOL).sum()), e2.size)
    check(out2, ref, c[2])


def _structures():
    """(name, gridpp_amd constructor, oracle Struct constructor): every structure function besides the plain Barnes one"""
    return [
        ("cressman", lambda g: g.CressmanStructure(30000, 300, 0.7), lambda O: O.Struct("Cressman", 30000, 300, 0.7)),
        ("soar", lambda g: g.SoarStructure(6000, 200, 0.5), lambda O: O.Struct("Soar", 6000, 200, 0.5)),
        ("toar", lambda g: g.ToarStructure(5000, 150, 0.4), lambda O: O.Struct("Toar", 5000, 150, 0.4)),
        ("powerlaw", lambda g: g.PowerlawStructure(4000, 1, 0.7), lambda O: O.Struct("Powerlaw", 4000, 1, 0.7)),
        # (the Linear kernel has no localization distance of its own, structure.cpp:765-789: it serves as a vertical / laf factor)
        ("multiple_linear", lambda g: g.MultipleStructure(g.BarnesStructure(15000), g.LinearStructure(0, 0.2, 0), g.PowerlawStructure(1, 1, 0.7)),
         lambda O: O.Struct.multiple(O.Struct("Barnes", 15000), O.Struct("Linear", 0, 0.2, 0), O.Struct("Powerlaw", 1, 1, 0.7))),
        ("multiple_nested", lambda g: g.MultipleStructure(g.SoarStructure(6000), g.MultipleStructure(g.BarnesStructure(1), g.CressmanStructure(1, 400, 1), g.BarnesStructure(1)), g.BarnesStructure(1, 1, 0.6)),
         lambda O: O.Struct.multiple(O.Struct("Soar", 6000), O.Struct.multiple(O.Struct("Barnes", 1), O.Struct("Cressman", 1, 400, 1), O.Struct("Barnes", 1)), O.Struct("Barnes", 1, 1, 0.6))),
        ("crossvalidation", lambda g: g.CrossValidation(g.BarnesStructure(15000, 200, 0.5), 4000),
         lambda O: O.Struct("Barnes", 15000, 200, 0.5).cross_validation(4000)),
    ]


@pytest.mark.parametrize("which", [s[0] for s in _structures()])
@pytest.mark.parametrize("max_points", [12, 0])
def test_ensi_other_structure_functions(which, max_points):
    """EnSI with Cressman / SOAR / TOAR / Powerlaw kernels, MultipleStructures (a Linear vertical factor, a nested one) and a CrossValidation wrapper
    (oi_ensi.cpp:213,250: localization_distance(p1) and corr_background(p1, p2) of ANY structure; structure.cpp:287-944),
    the 32-row tile path (max_points 12) and the large-n kernel (max_points 0), against the oracle."""
    import gridpp_amd as gridpp
    from oracle import oracle as O
    _, mk_g, mk_o = [s for s in _structures() if s[0] == which][0]
    lats, lons, bg, plat, plon, pbg, obs, sig = case(700 + len(which), 18, 16, 12, 90)
    Y, X, E = bg.shape
    rng = np.random.default_rng(5)
    ge, gl = rng.uniform(0, 500, (Y, X)), rng.uniform(0, 1, (Y, X))
    pe, pl = rng.uniform(0, 500, plat.size), rng.uniform(0, 1, plat.size)
    grid, points = gridpp.Grid(lats, lons, ge, gl), gridpp.Points(plat, plon, pe, pl)
    out = np.asarray(gridpp.optimal_interpolation_ensi(grid, bg, points, obs, sig, pbg, mk_g(gridpp), max_points))
    og, op = O.Pts(lats.ravel(), lons.ravel(), ge.ravel(), gl.ravel()), O.Pts(plat, plon, pe, pl)
    ref = O.oi_ensi_generic(og, bg.reshape(-1, E), op, obs, sig, pbg, mk_o(O), max_points).reshape(Y, X, E)
    check(out, ref, bg)


@pytest.mark.parametrize("kind", ["Barnes", "Soar"])
@pytest.mark.parametrize("max_points", [10, 0])
def test_ensi_spatially_varying_scales_against_the_oracle(kind, max_points):
    """Spatially varying h / v / w on a coarser field grid (structure.cpp:168-214): the scales (and the localization distance)
    of a grid point are those of its nearest field point; oracle = the same lookup + the structure as seen from the grid point."""
    import gridpp_amd as gridpp
    from oracle import oracle as O
    lats, lons, bg, plat, plon, pbg, obs, sig = case(811, 20, 22, 10, 120)
    Y, X, E = bg.shape
    rng = np.random.default_rng(6)
    ge, pe = rng.uniform(0, 500, (Y, X)), rng.uniform(0, 500, plat.size)
    flat_, flon_ = np.meshgrid(np.linspace(0, 1, 7), np.linspace(0, 1, 9), indexing="ij")
    base = {"Barnes": 14000, "Soar": 4000}[kind]
    hf = (base * rng.uniform(0.7, 1.3, flat_.shape)).astype(np.float32)
    vf = (300 * rng.uniform(0.7, 1.3, flat_.shape)).astype(np.float32)
    wf = np.zeros(flat_.shape, np.float32)
    grid, points, fgrid = gridpp.Grid(lats, lons, ge, ()), gridpp.Points(plat, plon, pe, ()), gridpp.Grid(flat_, flon_)
    st = getattr(gridpp, kind + "Structure")(fgrid, hf, vf, wf, 0.0013)
    out = np.asarray(gridpp.optimal_interpolation_ensi(grid, bg, points, obs, sig, pbg, st, max_points))
    og, op, of = O.Pts(lats.ravel(), lons.ravel(), ge.ravel()), O.Pts(plat, plon, pe), O.Pts(flat_.ravel(), flon_.ravel())
    ci = O.nearest_indices(of, og)
    Rf = np.array([O.structure_localization(kind, h, 0.0013) for h in hf.ravel()], np.float32)
    cp = [a.ravel()[ci] for a in (hf, vf, wf)] + [Rf[ci]]
    ref = O.oi_ensi_generic(og, bg.reshape(-1, E), op, obs, sig, pbg, O.Struct(kind, base), max_points, True, cp).reshape(Y, X, E)
    check(out, ref, bg)


@pytest.mark.parametrize("E,max_points", [(3, 5), (10, 10), (10, 30), (50, 30)])
def test_ensi_matches_oracle(E, max_points):
    c = case(100 + E, 24, 20, E, 60)
    out, ref = run(c, 20000, max_points)
    check(out, ref, c[2])


@pytest.mark.parametrize("E", [15, 16, 17, 20, 21, 32, 33, 36, 37, 47, 48, 49, 50, 51, 52, 53, 63, 64, 65, 66, 80, 100, 130])
@pytest.mark.parametrize("allow", [True, False])
def test_ensi_member_counts_around_the_tiles_of_sixteen(E, allow):
    """k_ensi_members covers the members in tiles of 16 on the matrix cores and takes up to four members beyond the last full tile on
    the vector unit (17..20, 33..36, 49..52 valid members): every tile count with a tail of 1..4, without one, and with the padded form
    (5..15 beyond a full tile), and more than 64 members (member chunks of 64, the table form of the update: `k_ensi_members<false>`);
    one member invalid somewhere for E = 51 / 53 / 66 so that the valid count differs from E"""
    c = case(300 + E, 12, 11, E, 40, nan_member=1 if E in (51, 53, 66) else None)
    out, ref = run(c, 20000, 30, allow=allow)
    check(out, ref, c[2])
    assert plain_err(out, ref).max() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("E", [16, 20, 33, 47, 50, 52, 64])
@pytest.mark.parametrize("allow", [True, False])
def test_ensi_two_area_member_kernel_behind_its_switch(E, allow, monkeypatch):
    """Round 6: up to 64 valid members go through `k_ensi_members3` (one staging area, U and Y in matrix-core operand layout in registers:
    three waves per SIMD, ensi_members3.h).  The kernel it replaced stays behind GPP_ENSI_MEMBERS2: same oracle, same measure, and the two
    kernels within float32 rounding of each other (they order some double-precision sums differently)."""
    c = case(300 + E, 12, 11, E, 40, nan_member=1 if E == 52 else None)
    new, ref = run(c, 20000, 30, allow=allow)
    monkeypatch.setenv("GPP_ENSI_MEMBERS2", "1")
    old, _ = run(c, 20000, 30, allow=allow)
    check(old, ref, c[2])
    assert plain_err(old, ref).max() < 1e-5 and plain_err(new, ref).max() < 1e-5
    assert plain_err(new, old).max() < 1e-5


def test_ensi_no_extrapolation_nan_obs_invalid_member():
    c = case(7, 20, 24, 8, 50, nan_member=2, nan_obs=True)
    out, ref = run(c, 20000, 12, allow=False)
    check(out, ref, c[2])
    # the member that is invalid somewhere is untouched everywhere (src/api/oi_ensi.cpp:187-201)
    bg = c[2]
    m = ~np.isnan(bg[:, :, 2])
    assert (out[:, :, 2][m] == bg[:, :, 2][m]).all()


def test_ensi_elev_structure_and_points_overload():
    import gridpp_amd as gridpp
    from oracle import oracle as O
    c = case(9, 16, 16, 6, 40)
    out, ref = run(c, 25000, 8, v=300, elev=True)
    check(out, ref, c[2])
    # Points overload: background (N, E)
    lats, lons, bg, plat, plon, pbg, obs, sig = c
    n = 300
    rng = np.random.default_rng(5)
    blat, blon = rng.random(n), rng.random(n)
    b2 = rng.normal(0, 1, (n, 6)).astype(np.float32)
    out = gridpp.optimal_interpolation_ensi(gridpp.Points(blat, blon), b2, gridpp.Points(plat, plon), obs, sig, pbg,
                                            gridpp.BarnesStructure(25000), 8)
    ref = O.oi_ensi(O.Pts(blat, blon), b2, O.Pts(plat, plon), obs, sig, pbg, O.Barnes(25000), 8)
    check(np.asarray(out), ref, b2)


@pytest.mark.parametrize("E,max_points,S,h", [(10, 0, 150, 30000), (50, 50, 200, 30000), (20, 45, 200, 25000), (64, 0, 120, 40000)])
def test_ensi_more_than_32_observations(E, max_points, S, h, monkeypatch):
    """max_points == 0 or > 32: grid points with more than 32 usable observations are solved by k_ensi_big (E x E
    formulation, one workgroup per cell); the others stay on the 32-row tile."""
    c = case(300 + E + max_points, 18, 16, E, S)
    out, ref = run(c, h, max_points)
    check(out, ref, c[2])
    out2, ref2 = run(c, h, max_points, allow=False)
    check(out2, ref2, c[2])
    # the case does need the large-n kernel: without it the call fails loudly
    monkeypatch.setenv("GPP_ENSI_NO_BIG", "1")
    with pytest.raises(RuntimeError, match="more usable observations"):
        run(c, h, max_points)


def test_ensi_large_n_with_elevation_nan_obs_and_invalid_member():
    c = case(777, 14, 15, 12, 180, nan_member=5, nan_obs=True)
    out, ref = run(c, 30000, 0, allow=False, v=200, elev=True)
    check(out, ref, c[2])
    assert np.isnan(out[3, 4, 5])


def test_ensi_large_n_newton_schulz_kernel():
    """60 usable observations per grid point, 20 and 50 members: the 32-row tile path takes none of the cells, k_ensi_big_ns all of them
    (the cyclic-Jacobi kernel of round 1 it replaced left the library in round 4)."""
    for E in (20, 50):
        c = case(950 + E, 6, 7, E, 60)
        out, ref = run(c, 200000, 0)
        check(out, ref, c[2])


def test_passthrough_count_and_the_reference_warning(capsys):
    """oi_ensi.cpp:386-390,557-561: a grid point whose E x E system has rcond <= 0 keeps its background values and is counted; the count
    comes back through gpp_ensi_last_stats and the mirror prints the reference's warning.  With ONE valid member Pinv is the zero matrix:
    every grid point with an observation in range is such a point (and only those)."""
    import gridpp_amd as gridpp
    rng = np.random.default_rng(5)
    Y, X, E, S = 20, 24, 4, 30
    lats, lons = np.meshgrid(np.linspace(60, 60.5, Y), np.linspace(10, 11, X), indexing="ij")
    bg = rng.normal(0, 1, (Y, X, E)).astype(np.float32)
    bg[3, 4, 1:] = np.nan                       # members 1..3 are invalid somewhere: one valid member left (oi_ensi.cpp:187-201)
    plat, plon = 60.05 + 0.1 * rng.random(S), 10.1 + 0.2 * rng.random(S)      # a cluster in one corner: most grid points see no observation
    pbg = rng.normal(0, 1, (S, E)).astype(np.float32)
    obs, sig = rng.normal(0, 1, S).astype(np.float32), np.ones(S, np.float32)
    grid, points = gridpp.Grid(lats, lons), gridpp.Points(plat, plon)
    out = gridpp.optimal_interpolation_ensi(grid, bg, points, obs, sig, pbg, gridpp.BarnesStructure(3000), 10)
    assert np.array_equal(out, bg, equal_nan=True)
    st = gridpp.ensi_last_stats()
    in_range = int((gridpp.count(points, grid, gridpp.BarnesStructure(3000).localization_distance()) > 0).sum())
    assert st["cells"] == Y * X and 0 < st["condition_passthrough"] < Y * X
    assert st["condition_passthrough"] == in_range
    assert "Warning: Condition number error in %d points. Using raw values in those points." % st["condition_passthrough"] in capsys.readouterr().out
    # a healthy call: no count, no warning
    bg2 = rng.normal(0, 1, (Y, X, E)).astype(np.float32)
    out2 = gridpp.optimal_interpolation_ensi(grid, bg2, points, obs, sig, pbg, gridpp.BarnesStructure(3000), 10)
    assert gridpp.ensi_last_stats()["condition_passthrough"] == 0 and "Warning" not in capsys.readouterr().out
    assert np.abs(out2 - bg2).max() > 0


def test_ensi_large_n_limits():
    """No capacity limit (oi_ensi.cpp:187-201,244-261 have none): 480 usable observations per grid point (several chunks of Y in
    k_ensi_big), 600 observations with 80 valid members (beyond one member per lane: k_ensi_huge, everything in HBM scratch), and
    9 000 candidates per grid point (beyond the LDS sort of k_ensi_big) all return the oracle's values."""
    c = case(901, 5, 6, 16, 480)
    out, ref = run(c, 200000, 0)                     # every observation is in range of every cell
    check(out, ref, c[2])
    c2 = case(902, 4, 4, 80, 600)
    out, ref = run(c2, 200000, 0)
    check(out, ref, c2[2])
    out, ref = run(c2, 200000, 45, allow=False)      # the same ensemble with the top-45 cut and the anti-extrapolation clamp
    check(out, ref, c2[2])
    c3 = case(903, 2, 3, 6, 9000)
    out, ref = run(c3, 300000, 40)                   # 9 000 candidates, 40 kept
    check(out, ref, c3[2])


# ---- vectors of an independent LAPACK restatement + the analytic 1-observation update (tests/golden/ensi_cases.npz,
# tools/make_ensi_fixtures.py): the same vectors pin the oracle on the CPU (tests/test_oracle_golden.py)
from tests import ensi_golden  # noqa: E402


@pytest.mark.parametrize("name", ensi_golden.NAMES)
def test_ensi_golden_vectors(name):
    import gridpp_amd as gridpp
    c = ensi_golden.CASES[name]
    h, v, w, mp, allow = c["params"]
    nb, ns = c["blat"].size, c["plat"].size
    belev, blaf = c.get("belev", np.full(nb, np.nan, np.float32)), c.get("blaf", np.full(nb, np.nan, np.float32))
    pelev, plaf = c.get("pelev", np.full(ns, np.nan, np.float32)), c.get("plaf", np.full(ns, np.nan, np.float32))
    points = gridpp.Points(c["plat"], c["plon"], pelev, plaf)
    E = c["background"].shape[1]
    if "shape" in c and c["shape"][0] > 0:      # Grid overload
        Y, X = int(c["shape"][0]), int(c["shape"][1])
        grid = gridpp.Grid(c["blat"].reshape(Y, X), c["blon"].reshape(Y, X), belev.reshape(Y, X), blaf.reshape(Y, X))
        if "hfield" in c:   # spatially varying scales on the background grid
            st = gridpp.BarnesStructure(grid, c["hfield"].reshape(Y, X), c["vfield"].reshape(Y, X), c["wfield"].reshape(Y, X))
        else:
            st = gridpp.BarnesStructure(h, v, w)
        out = gridpp.optimal_interpolation_ensi(grid, c["background"].reshape(Y, X, E), points, c["pobs"], c["psigmas"],
                                                c["pbackground"], st, int(mp), bool(allow))
    else:                                        # Points overload
        bpoints = gridpp.Points(c["blat"], c["blon"], belev, blaf)
        out = gridpp.optimal_interpolation_ensi(bpoints, c["background"], points, c["pobs"], c["psigmas"], c["pbackground"],
                                                gridpp.BarnesStructure(h, v, w), int(mp), bool(allow))
    out = np.asarray(out)
    assert out.dtype == np.float32
    ensi_golden.check(out.reshape(nb, E), c)


def test_ensi_in_many_batches_of_tiles(monkeypatch):
    """The spectral kernel and the ensemble kernel run in batches of tiles (what the second needs of a cell waits in an HBM park of
    bounded size): one tile per batch here must give the values of the single-batch run, bit for bit."""
    import gridpp_amd as gridpp
    c = ensi_golden.CASES["e50_mp30_noextrap"]
    h, v, w, mp, allow = c["params"]
    Y, X = int(c["shape"][0]), int(c["shape"][1])
    E = c["background"].shape[1]
    grid = gridpp.Grid(c["blat"].reshape(Y, X), c["blon"].reshape(Y, X), c["belev"].reshape(Y, X), c["blaf"].reshape(Y, X))
    points = gridpp.Points(c["plat"], c["plon"], c["pelev"], c["plaf"])
    args = (grid, c["background"].reshape(Y, X, E), points, c["pobs"], c["psigmas"], c["pbackground"], gridpp.BarnesStructure(h, v, w), int(mp), bool(allow))
    one = np.asarray(gridpp.optimal_interpolation_ensi(*args))
    monkeypatch.setenv("GPP_ENSI_PARK_MB", "1")
    many = np.asarray(gridpp.optimal_interpolation_ensi(*args))
    assert np.array_equal(one, many, equal_nan=True)
    ensi_golden.check(one, c)


@pytest.mark.parametrize("E,mp,S", [(50, 30, 150), (20, 10, 100), (30, 50, 150), (70, 40, 120)])
@pytest.mark.parametrize("kw", [dict(sig_scale=0.1), dict(sig_scale=0.01), dict(spread=0.01), dict(spread=100.0), dict(sig_scale=0.1, spread=0.01),
                                dict(offset=20.0), dict(dup=True), dict(sig_scale=0.01, offset=20.0, dup=True)], ids=lambda kw: ",".join("%s=%g" % kv for kv in kw.items()))
def test_ensi_off_the_benign_manifold(E, mp, S, kw):
    """Round 5 (tools/ensi_illcond.py, profiles/r05_ensi_illcond.txt): observation sigmas far below the ensemble spread (large Y^T R^-1 Y, many
    sweeps, a large diagonal against the fixed 0.040 c cut of the default mode), a spread far from the observation error, observations 20
    spreads away, two near-duplicate members -- on the tile path (max_points <= 32, both modes) and on the large-n kernels, in the PLAIN measure.
    (E = 30, max_points = 50 with sigma x 0.01 is the case whose Newton-Schulz iteration did not converge until round 5, E = 70 the one whose
    Jacobi stopped at 1e-11 * trace: /root/reference/src/api/oi_ensi.cpp:379-437 is all float64.)  The axis ends where the reference stops
    reproducing itself: with spread / sigma >= 1e3 two faithful implementations of its inv + eig_sym (the oracle and the LAPACK restatement)
    differ by 1e-5 and more (tools/ensi_illcond.py prints that column; profiles/r05_ensi_illcond.txt)."""
    import gridpp_amd as gridpp
    from oracle import oracle as O
    lats, lons, bg, plat, plon, pbg, obs, sig = case(4100 + E, 6, 6, E, S)
    sig = (sig * kw.get("sig_scale", 1.0)).astype(np.float32)
    sp = kw.get("spread", 1.0)
    if sp != 1.0:
        m = bg.mean(axis=2, keepdims=True); bg = (m + sp * (bg - m)).astype(np.float32)
        pm = pbg.mean(axis=1, keepdims=True); pbg = (pm + sp * (pbg - pm)).astype(np.float32)
    if kw.get("offset"):
        obs = (obs + kw["offset"] * np.where(np.arange(S) % 2, -1, 1)).astype(np.float32)
    if kw.get("dup"):
        bg[:, :, E - 1] = bg[:, :, 0] * np.float32(1 + 1e-6); pbg[:, E - 1] = pbg[:, 0] * np.float32(1 + 1e-6)
    h = 60000.0
    ref = O.oi_ensi(O.Pts(lats.ravel(), lons.ravel()), bg.reshape(-1, E), O.Pts(plat, plon), obs, sig, pbg, O.Barnes(h), mp, True).reshape(bg.shape)
    for converged in (False, True):
        gridpp.ensi_set_convergence(converged)
        try:
            out = np.asarray(gridpp.optimal_interpolation_ensi(gridpp.Grid(lats, lons), bg, gridpp.Points(plat, plon), obs, sig, pbg, gridpp.BarnesStructure(h), mp, True))
        finally:
            gridpp.ensi_set_convergence(False)
        assert (np.isnan(out) == np.isnan(ref)).all()
        err = plain_err(out, ref)
        assert err.max() < RTOL, (converged, float(err.max()))
