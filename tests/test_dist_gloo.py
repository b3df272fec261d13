"""N > 1 path on CPU: two gloo ranks, row tiles + observation broadcast (gridpp_amd/dist.py), with the single-GPU
compute stage replaced by the oracle.  The assembled tiles must equal the single-process result bit for bit."""
import os
import socket

import numpy as np
import pytest

from gridpp_amd import dist as gdist


def test_row_tiles_cover_the_grid():
    for ny in (1, 7, 8, 4000, 4001):
        for world in (1, 2, 3, 8):
            tiles = gdist.all_tiles(ny, world)
            assert tiles[0][0] == 0 and tiles[-1][1] == ny
            assert all(a[1] == b[0] for a, b in zip(tiles, tiles[1:]))
            sizes = [b - a for a, b in tiles]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        gdist.row_tile(10, 2, 2)


def _oracle_compute(lats, lons, bg, plats, plons, obs, ratios, pbg, structure_args, max_points, allow=True):
    from oracle import oracle as O
    g = O.Pts(np.ravel(lats), np.ravel(lons))
    p = O.Pts(plats, plons)
    return O.oi(g, np.ravel(bg), p, obs, ratios, pbg, O.Barnes(*structure_args), max_points, allow).reshape(np.shape(bg))


def _workload():
    rng = np.random.default_rng(77)
    Y, X, S = 21, 16, 60
    lats, lons = np.meshgrid(np.linspace(0, 1, Y), np.linspace(0, 1, X), indexing="ij")
    bg = rng.normal(0, 1, (Y, X)).astype(np.float32)
    plats, plons = rng.random(S), rng.random(S)
    vals = np.stack([rng.normal(0, 1, S), rng.uniform(0.1, 1, S), rng.normal(0, 1, S)]).astype(np.float32)
    return lats, lons, bg, plats, plons, vals


def _worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lats, lons, bg, plats, plons, vals = _workload()
    # only rank 0 owns the observation values of this step; the others start from garbage
    t = torch.from_numpy(vals.copy()) if rank == 0 else torch.full((3, vals.shape[1]), float("nan"))
    row0, row1, tile = gdist.tiled_optimal_interpolation(lats, lons, bg, plats, plons, t, (30000.0,), 8, rank, world,
                                                         _oracle_compute)
    assert torch.equal(t, torch.from_numpy(vals)), "broadcast did not deliver the observation block"
    np.save(os.path.join(outdir, "tile%d.npy" % rank), tile)
    np.save(os.path.join(outdir, "rows%d.npy" % rank), np.array([row0, row1]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_tiles_equal_single_process(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 2
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    lats, lons, bg, plats, plons, vals = _workload()
    ref = _oracle_compute(lats, lons, bg, plats, plons, vals[0], vals[1], vals[2], (30000.0,), 8)
    out = np.full_like(ref, np.nan)
    for r in range(world):
        row0, row1 = np.load(tmp_path / ("rows%d.npy" % r))
        out[row0:row1] = np.load(tmp_path / ("tile%d.npy" % r))
    np.testing.assert_array_equal(out, ref)
    assert np.abs(ref - bg).max() > 0.05


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("hw", [0, 2, 7, 40])
def test_neighbourhood_halo_tiles_equal_single_call(world, hw):
    """Row tiles + halfwidth-row halo reproduce the single-call neighbourhood exactly (oracle as the compute stage)."""
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    f = rng.uniform(0, 10, (37, 23)).astype(np.float32)
    f[4:7, 3:9] = np.nan
    for stat in (O.Mean, O.Max, O.Count):
        ref = O.neighbourhood(f, hw, stat)
        out = np.full_like(ref, -1)
        for r in range(world):
            row0, row1, tile = gdist.tiled_neighbourhood(f, hw, stat, r, world, O.neighbourhood)
            out[row0:row1] = tile
        np.testing.assert_array_equal(out, ref)


def _stream_worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S, steps = 17, 7
    slots = [torch.full((3, S), float("nan")) for _ in range(2)]

    def fill(slot, step):                       # rank 0 produces a new block per step
        slots[slot].copy_(torch.arange(3 * S, dtype=torch.float32).reshape(3, S) + 1000.0 * step)

    stream = gdist.ObservationStream(slots, rank, fill)
    seen = []
    for k in range(steps):
        v = stream.next()
        seen.append(v.clone().numpy())          # "compute" of step k reads the block while k+1 is in flight
    stream.drain()
    assert stream.pending == [None, None]
    np.save(os.path.join(outdir, "seen%d.npy" % rank), np.stack(seen))
    dist.barrier()
    dist.destroy_process_group()


def test_double_buffered_observation_stream_two_ranks(tmp_path):
    """Every rank sees step k's block in step k, although block k+1 is posted before step k is consumed."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_stream_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    want = np.stack([np.arange(51, dtype=np.float32).reshape(3, 17) + 1000.0 * k for k in range(7)])
    for r in range(2):
        np.testing.assert_array_equal(np.load(tmp_path / ("seen%d.npy" % r)), want)


def test_observation_stream_single_process():
    import torch
    slots = [torch.zeros(2), torch.zeros(2)]
    stream = gdist.ObservationStream(slots, 0, lambda slot, step: slots[slot].fill_(step))
    assert [float(stream.next()[0]) for _ in range(4)] == [0.0, 1.0, 2.0, 3.0]
    stream.drain()


def _halo_worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(11)
    full = rng.uniform(0, 10, (40, 13, 3)).astype(np.float32)      # every rank can regenerate the full field: only its rows are used
    hw = 4
    row0, row1 = gdist.row_tile(full.shape[0], rank, world)
    halo = gdist.HaloExchange(torch.from_numpy(full[row0:row1].copy()), hw, rank, world)
    for step in range(2):                                         # the second exchange must leave the tile rows untouched
        padded, top = halo.exchange()
    lo, hi, _, _ = gdist.halo_rows(full.shape[0], rank, world, hw)
    np.testing.assert_array_equal(padded.numpy(), full[lo:hi])    # tile + the neighbours' rows = the rows a single process would see
    out = O.neighbourhood(padded.numpy(), hw, O.Mean)[top:top + (row1 - row0)]
    np.save(os.path.join(outdir, "nb%d.npy" % rank), out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_device_side_halo_exchange(tmp_path, world):
    """HaloExchange (send / recv of `halfwidth` rows between neighbouring ranks) + the filter on the padded tile = the single call."""
    import torch.multiprocessing as mp
    from oracle import oracle as O
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_halo_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(11)
    full = rng.uniform(0, 10, (40, 13, 3)).astype(np.float32)
    ref = O.neighbourhood(full, 4, O.Mean)
    got = np.concatenate([np.load(tmp_path / ("nb%d.npy" % r)) for r in range(world)])
    np.testing.assert_array_equal(got, ref)


def test_weighted_row_tiles_partition_and_balance():
    """weighted_row_tiles: contiguous, covering, every tile about 1 / world of the total weight; equal weights give the equal tiles;
    degenerate inputs (zero weights, fewer rows than ranks) stay valid partitions."""
    from gridpp_amd import dist as gdist
    rng = np.random.default_rng(3)
    for world in (1, 2, 3, 8):
        for ny in (1, 5, 64, 1000):
            w = rng.uniform(0.2, 3.0, ny)
            tiles = gdist.weighted_row_tiles(w, world)
            assert len(tiles) == world and tiles[0][0] == 0 and tiles[-1][1] == ny
            assert all(tiles[r][1] == tiles[r + 1][0] for r in range(world - 1)) and all(a <= b for a, b in tiles)
            if ny >= 50 * world:
                share = np.array([w[a:b].sum() for a, b in tiles]) / w.sum()
                assert np.abs(share - 1.0 / world).max() < 3.0 * w.max() / w.sum() + 1e-12
            assert gdist.weighted_row_tiles(np.ones(ny), world) == [(min(a, ny), min(b, ny)) for a, b in gdist.weighted_row_tiles(np.ones(ny), world)]
    assert gdist.weighted_row_tiles(np.ones(1000), 8) == gdist.all_tiles(1000, 8)
    assert gdist.weighted_row_tiles(np.zeros(10), 4) == gdist.all_tiles(10, 4)
    # a cost model: rows near the observations cost more -> the tiles there are shorter
    lat = np.linspace(0, 1, 400)
    cost = gdist.row_cost_from_observations(lat, 0.2 + 0.1 * rng.random(500), 0.05, base=1.0, per_obs=0.05)
    tiles = gdist.weighted_row_tiles(cost, 4)
    heights = [b - a for a, b in tiles]
    assert min(heights) < 100 < max(heights) and sum(heights) == 400


class _LazyAnalysis:
    """a deferred analysis that reads its observation block as LATE as the contract allows: at wait()"""
    def __init__(self, view, fn):
        self.view, self.fn = view, fn

    def wait(self):
        return self.fn(self.view.clone().numpy())


def _pipeline_worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lats, lons, bg, plats, plons, vals = _workload()
    steps, ahead = 6, 1
    row0, row1 = gdist.row_tile(lats.shape[0], rank, world)
    slots = [torch.full((3, vals.shape[1]), float("nan")) for _ in range(ahead + 2)]

    def fill(slot, step):                       # rank 0: a new observation set per step
        slots[slot].copy_(torch.from_numpy(vals) + float(step))

    def compute(v):
        return _oracle_compute(lats[row0:row1], lons[row0:row1], bg[row0:row1], plats, plons, v[0], v[1], v[2], (30000.0,), 8)
    stream, pipe, tiles = gdist.ObservationStream(slots, rank, fill), gdist.AnalysisPipeline(ahead), []
    for k in range(steps):
        v = stream.next()
        r = pipe.push(_LazyAnalysis(v, compute))          # the block of step k is read when step k is WAITED for: after block k+1 was posted
        if r is not None:
            tiles.append(r)
    tiles += pipe.drain()
    stream.drain()
    assert len(tiles) == steps
    np.save(os.path.join(outdir, "ptiles%d.npy" % rank), np.stack(tiles))
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_analyses_equal_the_synchronous_loop(tmp_path):
    """gridpp_amd.dist.AnalysisPipeline + a three-slot ObservationStream (what bench.py --gpus N does with optimal_interpolation_async): the
    analysis of step k is completed one step late, its observation block must still be the block of step k -- every tile of every step equals
    the single-process, one-at-a-time result bit for bit."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_pipeline_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    lats, lons, bg, plats, plons, vals = _workload()
    got = np.concatenate([np.load(tmp_path / ("ptiles%d.npy" % r)) for r in range(2)], axis=1)
    for k in range(6):
        v = vals + np.float32(k)
        np.testing.assert_array_equal(got[k], _oracle_compute(lats, lons, bg, plats, plons, v[0], v[1], v[2], (30000.0,), 8))


def test_analysis_pipeline_order_and_depth():
    class P:
        def __init__(self, i, log): self.i, self.log = i, log
        def wait(self): self.log.append(self.i); return self.i
    for ahead in (0, 1, 3):
        log, pipe, got = [], gdist.AnalysisPipeline(ahead), []
        for i in range(6):
            r = pipe.push(P(i, log))
            assert (r is None) == (i < ahead)
            if r is not None:
                got.append(r)
            assert len(pipe.q) == min(i + 1, ahead)
        got += pipe.drain()
        assert got == log == list(range(6))
    with pytest.raises(ValueError):
        gdist.ObservationStream([None], 0)
