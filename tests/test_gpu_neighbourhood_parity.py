"""HIP neighbourhood filters vs the CPU oracle on seeded inputs (through the C-ABI).
Tolerance: 1e-5 relative (BASELINE.json north_star); counts and min/max are exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def field(seed, Y, X, E=None, nan_blocks=True):
    rng = np.random.default_rng(seed)
    shape = (Y, X) if E is None else (Y, X, E)
    f = rng.uniform(0, 10, shape).astype(np.float32)
    if nan_blocks:
        f[3:9, 5:20] = np.nan
        f[rng.random(shape) < 0.03] = np.nan
        if E is not None:
            f[20:24, 30:36, :] = np.nan   # cells with no valid member
    return f


def close(a, b, exact=False):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == np.float32
    assert (np.isnan(a) == np.isnan(b)).all(), (np.isnan(a).sum(), np.isnan(b).sum())
    m = ~np.isnan(b)
    if not m.any():
        return
    if exact:
        assert (a[m] == b[m]).all()
    else:
        err = np.abs(a[m].astype(np.float64) - b[m]) / np.maximum(np.abs(b[m]), 1e-3)
        assert err.max() < RTOL, err.max()


@pytest.fixture(scope="module")
def api():
    import gridpp_amd as gridpp
    from oracle import oracle as O
    return gridpp, O


@pytest.mark.parametrize("hw", [0, 1, 7, 15, 300])
def test_2d_statistics(api, hw):
    gridpp, O = api
    f = field(1, 257, 193)
    for stat in (gridpp.Mean, gridpp.Sum, gridpp.Std, gridpp.Variance):
        out = gridpp.neighbourhood(f, hw, stat)
        ref = O.neighbourhood(f, hw, stat)
        if stat in (gridpp.Std, gridpp.Variance):
            # Variance = E[x^2] - E[x]^2 from two float32 box means (neighbourhood.cpp:211-235, not clamped).  The box means
            # agree with the reference's summed-area table to a float32 ulp, so the variance agrees to ~3 ulp of E[x^2]:
            # 1e-5 RELATIVE holds wherever the subtraction does not cancel (variance >= 5 % of E[x^2]); the cells outside
            # 1e-5 are counted and must all be cancelling ones, where parity is 1e-5 of the cancelling terms instead.
            both = ~np.isnan(ref) & ~np.isnan(out)
            assert both.sum() > 0.9 * (~np.isnan(f)).sum()
            m2 = O.neighbourhood(np.where(np.isnan(f), np.nan, f * f).astype(np.float32), hw, gridpp.Mean)
            var_ref = O.neighbourhood(f, hw, gridpp.Variance)
            cancelling = ~(var_ref >= 0.05 * m2)
            rel = np.abs(out.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-30)
            outside = both & (rel > RTOL)
            assert not (outside & ~cancelling).any(), ("non-cancelling cells outside 1e-5", int((outside & ~cancelling).sum()))
            assert (both & ~cancelling).sum() > 0.5 * both.sum() or hw == 0       # the tight bound covers most of the field
            scale = m2 if stat == gridpp.Variance else np.sqrt(np.maximum(m2, 0))
            sel = both & cancelling & (np.sign(out) == np.sign(ref) if stat == gridpp.Std else True)
            if stat == gridpp.Variance:
                # (+ the absolute noise of the reference's summed-area table: its double prefix sums run over the whole field, ~1e-16 * sum|f|
                #  = 3e-11 here, which shows where E[x^2] itself is tiny)
                assert (np.abs(out[sel] - ref[sel]) <= 1e-5 * scale[sel] + 1e-9).all()
            # (a cancelling Std is the square root of a difference of ~1e-7 * E[x^2]: only its NaN-ness / order of magnitude is defined)
        else:
            close(out, ref)
    for stat in (gridpp.Count, gridpp.Min, gridpp.Max):
        close(gridpp.neighbourhood(f, hw, stat), O.neighbourhood(f, hw, stat), exact=True)


def test_more_rows_than_one_launch_dimension_holds(api):
    """Grids with more than 65 535 rows (gridDim.y is a 16-bit quantity): the row passes run in row chunks."""
    gridpp, O = api
    f = field(7, 70001, 9)
    for stat in (gridpp.Mean, gridpp.Sum):
        close(gridpp.neighbourhood(f, 2, stat), O.neighbourhood(f, 2, stat))
    f = field(8, 262150, 5)      # beyond 4 x 65 535 rows: the min / max passes chunk too
    for stat in (gridpp.Count, gridpp.Min, gridpp.Max):
        close(gridpp.neighbourhood(f, 2, stat), O.neighbourhood(f, 2, stat), exact=True)


@pytest.mark.parametrize("hw", [3, 40, 500])
def test_row_pass_for_windows_wider_than_an_lds_tile(api, hw, monkeypatch):
    """The sliding row pass that takes over when min(halfwidth, X) is beyond ~20 000 columns (forced here on a small field)."""
    gridpp, O = api
    monkeypatch.setenv("GPP_BOX_ROWS_WIDE", "1")
    f = field(9, 61, 333)
    for stat in (gridpp.Mean, gridpp.Sum):
        close(gridpp.neighbourhood(f, hw, stat), O.neighbourhood(f, hw, stat))
    close(gridpp.neighbourhood(f, hw, gridpp.Count), O.neighbourhood(f, hw, gridpp.Count), exact=True)
    g3 = field(10, 40, 70, 6)
    close(gridpp.neighbourhood(g3, hw, gridpp.Mean), O.neighbourhood(g3, hw, gridpp.Mean))


@pytest.mark.parametrize("E", [1, 5, 100, 130])
def test_3d_statistics(api, E):
    gridpp, O = api
    f = field(2 + E, 70, 90, E)
    for stat in (gridpp.Mean, gridpp.Sum):
        close(gridpp.neighbourhood(f, 3, stat), O.neighbourhood(f, 3, stat))
    for stat in (gridpp.Count, gridpp.Min, gridpp.Max):
        close(gridpp.neighbourhood(f, 3, stat), O.neighbourhood(f, 3, stat), exact=True)


def test_median_and_brute_force(api):
    gridpp, O = api
    f = field(5, 40, 33)
    close(gridpp.neighbourhood(f, 2, gridpp.Median), O.neighbourhood(f, 2, O.Median), exact=True)
    for stat in (gridpp.Mean, gridpp.Sum, gridpp.Std, gridpp.Variance):
        close(gridpp.neighbourhood_brute_force(f, 2, stat), O.neighbourhood_brute_force(f, 2, stat), exact=True)
    for stat in (gridpp.Min, gridpp.Max, gridpp.Median, gridpp.Count):
        close(gridpp.neighbourhood_brute_force(f, 2, stat), O.neighbourhood_brute_force(f, 2, stat), exact=True)
    f3 = field(6, 20, 18, 4)
    close(gridpp.neighbourhood_brute_force(f3, 1, gridpp.Mean), O.neighbourhood_brute_force(f3, 1, O.Mean), exact=True)


@pytest.mark.parametrize("q", [0.0, 0.1, 0.5, 0.9, 1.0])
def test_exact_quantile(api, q):
    gridpp, O = api
    f = field(7, 64, 64)
    close(gridpp.neighbourhood_quantile(f, q, 3), O.neighbourhood_quantile(f, q, 3), exact=True)
    f3 = field(8, 24, 20, 5)
    close(gridpp.neighbourhood_quantile(f3, q, 2), O.neighbourhood_quantile(f3, q, 2), exact=True)


@pytest.mark.parametrize("T", [1, 2, 11, 100])
def test_quantile_fast(api, T):
    gridpp, O = api
    thr = np.linspace(0, 10, T).astype(np.float32)
    f2 = field(9, 120, 96)
    f3 = field(10, 60, 72, 20)
    rng = np.random.default_rng(3)
    for f in (f2, f3):
        Y, X = f.shape[:2]
        for q in (0.0, 0.5, 0.9, 1.0):
            out = gridpp.neighbourhood_quantile_fast(f, q, 4, thr)
            ref = O.neighbourhood_quantile_fast(f, [q], 4, thr)
            close(out, ref)
        qf = rng.random((Y, X)).astype(np.float32)
        qf[0, 0], qf[1, 1], qf[2, 2] = 0.0, 1.0, np.nan
        close(gridpp.neighbourhood_quantile_fast(f, qf, 4, thr), O.neighbourhood_quantile_fast(f, qf, 4, thr))


@pytest.mark.parametrize("kind", ["permuted", "duplicate", "clustered", "nonfinite", "integer_ties"])
def test_quantile_fast_threshold_lists(api, kind):
    """The count pass ranks a member among the DISTINCT FINITE thresholds (k_qf_lut / k_qf_count): lists in any order, with duplicates,
    with two thresholds in one bucket of the rank table or a non-finite one (both take the compare-per-threshold pass), and members
    that sit exactly on thresholds; several 256-column strips and 64-row segments of the box pass, rows with missing members."""
    gridpp, O = api
    rng = np.random.default_rng(21)
    f = rng.uniform(0, 10, (90, 300, 8)).astype(np.float32)   # (two strips of the box pass, two row segments)
    f[rng.random(f.shape) < 0.01] = np.nan
    f[70:73, 100:110, :] = np.nan
    thr = np.linspace(0, 10, 9).astype(np.float32)
    if kind == "permuted":
        thr = rng.permutation(thr)
    elif kind == "duplicate":
        thr[3] = thr[5]
    elif kind == "clustered":
        thr[4] = np.nextafter(thr[3], np.float32(np.inf))
    elif kind == "nonfinite":
        thr[2], thr[7] = np.inf, np.nan
    else:
        f = np.round(f).astype(np.float32)
    def same(a, b):   # (a non-finite threshold can be the answer: infinities must agree exactly, the rest to 1e-5)
        a, b = np.asarray(a), np.asarray(b)
        assert (np.isinf(a) == np.isinf(b)).all() and (a[np.isinf(b)] == b[np.isinf(b)]).all()
        close(np.where(np.isinf(b), np.float32(0), a), np.where(np.isinf(b), np.float32(0), b))
    for q, hw in ((0.5, 15), (0.9, 3), (0.0, 16), (1.0, 0)):
        same(gridpp.neighbourhood_quantile_fast(f, q, hw, thr), O.neighbourhood_quantile_fast(f, [q], hw, thr))
    g = f[:, :, :7].copy()   # rows that are no whole float4s: the compare-per-threshold pass feeds the same box pass
    same(gridpp.neighbourhood_quantile_fast(g, 0.5, 7, thr), O.neighbourhood_quantile_fast(g, [0.5], 7, thr))


def test_thresholds_match_oracle(api):
    gridpp, O = api
    f = field(11, 50, 60, 7)
    for num in (1, 2, 5, 11, 100):
        np.testing.assert_array_equal(gridpp.get_neighbourhood_thresholds(f, num), O.get_neighbourhood_thresholds(f, num))
    g = np.round(field(12, 30, 30, nan_blocks=False))   # many duplicates
    g[:20] = 0
    for num in (2, 3, 4, 7):
        np.testing.assert_array_equal(gridpp.get_neighbourhood_thresholds(g, num), O.get_neighbourhood_thresholds(g, num))


def test_device_resident_path(api):
    """torch CUDA tensors in, torch tensor out: same numbers as the host path."""
    import torch
    gridpp, O = api
    f = field(13, 64, 80, 10)
    d = torch.from_numpy(f).cuda()
    out = gridpp.neighbourhood(d, 5, gridpp.Mean)
    assert out.is_cuda
    np.testing.assert_array_equal(out.cpu().numpy(), gridpp.neighbourhood(f, 5, gridpp.Mean))
    thr = np.linspace(0, 10, 11).astype(np.float32)
    out = gridpp.neighbourhood_quantile_fast(d, 0.5, 5, thr)
    np.testing.assert_array_equal(out.cpu().numpy(), gridpp.neighbourhood_quantile_fast(f, 0.5, 5, thr))


@pytest.mark.parametrize("shape", [(1, 1), (5, 3), (31, 64), (33, 65), (97, 130), (64, 200), (700, 70)])
def test_fused_box_pass_against_the_two_pass_form(api, shape):
    """k_box_march (both passes of the box statistics in one kernel, halfwidth <= 16) against the row pass + column pass it replaces
    (GPP_BOX_TWO_PASS) and the oracle: every halfwidth it serves, fields smaller than a strip and with ragged last strips and chunks, chunks with and
    without missing / infinite values (the counted and the closed-form windows), Mean / Sum / Count, and several planes at once
    (Std, Variance: two planes).  The two forms add the same doubles in another order: equal to the last float32 bit except where a
    sum sits on a rounding boundary, so they are compared at 1e-6 and each with the oracle at 1e-5; counts are exact."""
    gridpp, O = api
    Y, X = shape
    rng = np.random.default_rng(Y * 1000 + X)
    clean = rng.uniform(-5, 10, (Y, X)).astype(np.float32)
    holes = clean.copy()
    holes[rng.random((Y, X)) < 0.05] = np.nan
    if Y > 8 and X > 8:
        holes[2:6, 1:7] = np.nan
        holes[Y // 2, X // 2] = np.inf
    for f in (clean, holes):
        for hw in list(range(0, 33)):     # (17 .. 32: Min / Max only -- k_minmax_march over strips of 32 columns; the box statistics keep their two passes there)
            for stat in ((gridpp.Mean, gridpp.Sum, gridpp.Count) if hw <= 16 else ()) + (gridpp.Min, gridpp.Max) + ((gridpp.Variance,) if hw in (3, 16) else ()):
                fused = gridpp.neighbourhood(f, hw, stat)
                gridpp.set_path_override("GPP_BOX_TWO_PASS", "1")
                try:
                    two = gridpp.neighbourhood(f, hw, stat)
                finally:
                    gridpp.set_path_override("GPP_BOX_TWO_PASS", None)
                assert (np.isnan(fused) == np.isnan(two)).all()
                m = ~np.isnan(two)
                if stat in (gridpp.Count, gridpp.Min, gridpp.Max):     # (k_minmax_march: extrema are exact in any order)
                    assert (fused[m] == two[m]).all()
                    close(fused, O.neighbourhood(f, hw, stat), exact=True)
                elif stat == gridpp.Variance:
                    assert np.allclose(fused[m], two[m], rtol=0, atol=2e-5 * 100.0)   # (|f| <= 10: E[x^2] <= 100)
                else:
                    if m.any():
                        assert (np.abs(fused[m].astype(np.float64) - two[m]) <= 1e-6 * np.maximum(np.abs(two[m]), 1e-3)).all(), (hw, stat)
                    close(fused, O.neighbourhood(f, hw, stat))


def test_fused_box_pass_three_dimensional_mean(api):
    gridpp, O = api
    f = field(7, 70, 90, E=6)
    for hw in (2, 16):
        close(gridpp.neighbourhood(f, hw, gridpp.Mean), O.neighbourhood(f, hw, gridpp.Mean))


@pytest.mark.parametrize("shape", [(700, 70), (97, 130), (300, 40), (129, 64)])
def test_fused_box_pass_marching_through_many_chunks(api, shape):
    """The same kernel with ONE row segment per strip (GPP_BM_FILL=1: a workgroup marches through all rows, as it does on the 4000-row
    fields of config 4 -- small fields are otherwise cut into segments of one chunk): the ring of row sums wraps every two chunks, chunks with
    and without missing values follow each other (counts from the ring beside counts in closed form), the last chunk is ragged.  Against the
    two-pass form and the oracle, as above."""
    gridpp, O = api
    Y, X = shape
    rng = np.random.default_rng(Y * 7 + X)
    f = rng.uniform(-5, 10, (Y, X)).astype(np.float32)
    # missing values in some bands of rows only: chunks (32 rows) with and without them alternate
    for y0 in range(10, Y, 75):
        f[y0:y0 + 3, rng.integers(0, X, 5)] = np.nan
    f[Y // 2, X // 3] = np.inf
    f[Y - 1, 0] = np.nan
    gridpp.set_path_override("GPP_BM_FILL", "1")
    try:
        for hw in (0, 1, 5, 8, 15, 16, 17, 24, 32):
            for stat in ((gridpp.Mean, gridpp.Sum, gridpp.Count) if hw <= 16 else ()) + (gridpp.Min, gridpp.Max):
                fused = gridpp.neighbourhood(f, hw, stat)
                gridpp.set_path_override("GPP_BOX_TWO_PASS", "1")
                try:
                    two = gridpp.neighbourhood(f, hw, stat)
                finally:
                    gridpp.set_path_override("GPP_BOX_TWO_PASS", None)
                assert (np.isnan(fused) == np.isnan(two)).all()
                m = ~np.isnan(two)
                exact = stat in (gridpp.Count, gridpp.Min, gridpp.Max)
                if exact:
                    assert (fused[m] == two[m]).all()
                else:
                    assert (np.abs(fused[m].astype(np.float64) - two[m]) <= 1e-6 * np.maximum(np.abs(two[m]), 1e-3)).all(), (hw, stat)
                close(fused, O.neighbourhood(f, hw, stat), exact=exact)
    finally:
        gridpp.set_path_override("GPP_BM_FILL", None)
