"""gpp_optimal_interpolation_full(..., GPP_MEM_DEVICE | GPP_ASYNC) + gpp_wait (include/gridpp_hip.h; round 5): a repeated analysis streams
through the GPU without a host round trip per call.  Every deferred call must return the bits of the synchronous call -- in the steady state,
when the status block asks for more work than was enqueued (the wait runs the call again), when the call is not in the steady state at all
(it runs synchronously and its wait returns at once) and when more calls are pushed than slots exist.  The reference has no state between
calls (/root/reference/src/api/oi.cpp:221-338), so any interleaving is legal."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(seed, Y=200, X=240, S=260, rough_rows=0):
    import torch
    import gridpp_amd as gridpp
    rng = np.random.default_rng(seed)
    lats, lons = np.meshgrid(np.linspace(60, 60 + Y / 240.0, Y), np.linspace(10, 10 + X / 120.0, X), indexing="ij")
    ge = np.full((Y, X), 100.0, np.float32); gl = np.full((Y, X), 0.5, np.float32)
    if rough_rows:
        ge[-rough_rows:] = rng.uniform(0, 1000, (rough_rows, X)); gl[-rough_rows:] = rng.uniform(0, 1, (rough_rows, X))
    plat, plon = 60 + Y / 240.0 * rng.random(S), 10 + X / 120.0 * rng.random(S)
    pe, pl = np.full(S, 100.0, np.float32), np.full(S, 0.5, np.float32)
    grid, points = gridpp.Grid(lats, lons, ge, gl), gridpp.Points(plat, plon, pe, pl)
    st = gridpp.BarnesStructure(8000.0, 200.0, 0.5)
    bg = torch.from_numpy(rng.normal(0, 2, (Y, X)).astype(np.float32)).cuda()
    sets = []
    for k in range(6):
        o = rng.normal(0, 2, S).astype(np.float32)
        if k == 3:
            o[rng.random(S) < 0.3] = np.nan          # other usable observations: other selections, tiles the memory does not hold
        sets.append([torch.from_numpy(a).cuda() for a in (o, rng.uniform(0.05, 2, S).astype(np.float32), rng.normal(0, 2, S).astype(np.float32))])
    return gridpp, grid, points, st, bg, sets


@pytest.mark.parametrize("rough_rows", [0, 10], ids=["nothing_declined", "remembered_list"])
def test_deferred_calls_equal_synchronous_calls(rough_rows):
    gridpp, grid, points, st, bg, sets = _setup(5 + rough_rows, rough_rows=rough_rows)
    sync = [gridpp.optimal_interpolation(grid, bg, points, o, r, p, st, 30).cpu().numpy() for o, r, p in sets]
    stats_sync = gridpp.oi_last_stats()
    # one at a time
    for k, (o, r, p) in enumerate(sets):
        pend = gridpp.optimal_interpolation_async(grid, bg, points, o, r, p, st, 30)
        out = pend.wait().cpu().numpy()
        assert np.array_equal(out, sync[k], equal_nan=True), k
        s = pend.stats()
        assert s["cells"] == stats_sync["cells"] and s["cells_updated"] > 0 and s["kernel_ms"] > 0
    # all in flight at once (more than the library has slots for: the surplus runs synchronously), waited in order
    pend = [gridpp.optimal_interpolation_async(grid, bg, points, o, r, p, st, 30) for o, r, p in sets]
    import ctypes
    n = ctypes.c_int(-1)
    assert gridpp._capi.lib().gpp_pending(ctypes.byref(n)) == 0 and n.value == len(sets)
    for k in (2, 0, 1, 5, 4, 3):                       # (waiting for a later one completes the earlier ones first)
        assert np.array_equal(pend[k].wait().cpu().numpy(), sync[k], equal_nan=True), k
    assert gridpp._capi.lib().gpp_pending(ctypes.byref(n)) == 0 and n.value == 0
    # a synchronous call between deferred ones
    p0 = gridpp.optimal_interpolation_async(grid, bg, points, *sets[0], st, 30)
    mid = gridpp.optimal_interpolation(grid, bg, points, *sets[1], st, 30).cpu().numpy()
    p2 = gridpp.optimal_interpolation_async(grid, bg, points, *sets[2], st, 30)
    assert np.array_equal(mid, sync[1], equal_nan=True)
    assert np.array_equal(p2.wait().cpu().numpy(), sync[2], equal_nan=True) and np.array_equal(p0.wait().cpu().numpy(), sync[0], equal_nan=True)


def test_first_call_of_a_geometry_and_errors():
    import ctypes
    gridpp, grid, points, st, bg, sets = _setup(9)
    lib = gridpp._capi.lib()
    assert lib.gpp_wait() == gridpp._capi.GPP_EINVAL                      # nothing pending
    pend = gridpp.optimal_interpolation_async(grid, bg, points, *sets[0], st, 30)     # no memory of this geometry yet: runs synchronously
    ref = gridpp.optimal_interpolation(grid, bg, points, *sets[0], st, 30).cpu().numpy()
    assert np.array_equal(pend.wait().cpu().numpy(), ref, equal_nan=True)
    with pytest.raises(ValueError):
        gridpp.optimal_interpolation_async(grid, bg.cpu().numpy(), points, *[t.cpu().numpy() for t in sets[0]], st, 30)   # host arrays
    with pytest.raises(ValueError):
        gridpp.optimal_interpolation_async(grid, bg, points, *sets[0], st, -1)
    n = ctypes.c_int(-1)
    assert lib.gpp_pending(ctypes.byref(n)) == 0 and n.value == 0


def test_pipeline_over_a_stream_of_observation_sets():
    """what bench.py --case oi does: one analysis ahead (gridpp_amd.dist.AnalysisPipeline)"""
    from gridpp_amd import dist as gdist
    gridpp, grid, points, st, bg, sets = _setup(11, rough_rows=10)
    sync = [gridpp.optimal_interpolation(grid, bg, points, o, r, p, st, 30).cpu().numpy() for o, r, p in sets]
    pipe, got = gdist.AnalysisPipeline(1), []
    for rep in range(3):
        for o, r, p in sets:
            res = pipe.push(gridpp.optimal_interpolation_async(grid, bg, points, o, r, p, st, 30))
            if res is not None:
                got.append(res.cpu().numpy())
    got += [t.cpu().numpy() for t in pipe.drain()]
    assert len(got) == 18
    for k, g in enumerate(got):
        assert np.array_equal(g, sync[k % 6], equal_nan=True), k


def test_deferred_full_form_with_variance():
    """optimal_interpolation_full (analysis + analysis variance, oi.cpp:138-412) deferred: the bits of the blocking call, with a Points background too"""
    import torch
    gridpp, grid, points, st, bg, sets = _setup(13, rough_rows=10)
    rng = np.random.default_rng(3)
    bvar = torch.from_numpy(rng.uniform(0.5, 2, tuple(bg.shape)).astype(np.float32)).cuda()
    S = sets[0][0].shape[0]
    ovar = torch.from_numpy(rng.uniform(0.1, 1, S).astype(np.float32)).cuda()
    bvp = torch.from_numpy(rng.uniform(0.5, 2, S).astype(np.float32)).cuda()
    sync = [gridpp.optimal_interpolation_full(grid, bg, bvar, points, o, ovar, p, bvp, st, 30) for o, r, p in sets[:3]]
    for rep in range(2):
        pend = [gridpp.optimal_interpolation_full_async(grid, bg, bvar, points, o, ovar, p, bvp, st, 30) for o, r, p in sets[:3]]
        for k, pd in enumerate(pend):
            a, v = pd.wait()
            assert np.array_equal(a.cpu().numpy(), sync[k][0].cpu().numpy(), equal_nan=True) and np.array_equal(v.cpu().numpy(), sync[k][1].cpu().numpy(), equal_nan=True), (rep, k)
    # a Points background (64 consecutive points per tile)
    Y, X = bg.shape
    lats, lons = np.meshgrid(np.linspace(60, 60 + Y / 240.0, Y), np.linspace(10, 10 + X / 120.0, X), indexing="ij")
    pts_bg = gridpp.Points(lats.ravel(), lons.ravel(), np.full(Y * X, 100.0, np.float32), np.full(Y * X, 0.5, np.float32))
    flat = bg.reshape(-1).contiguous()
    ref = gridpp.optimal_interpolation(pts_bg, flat, points, *sets[0], st, 30).cpu().numpy()
    for _ in range(3):
        out = gridpp.optimal_interpolation_async(pts_bg, flat, points, *sets[0], st, 30).wait().cpu().numpy()
        assert np.array_equal(out, ref, equal_nan=True)
