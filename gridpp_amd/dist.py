"""Multi-GPU layer of the OI path: one process per GPU, the output grid is cut into contiguous row
tiles, the observation values of each analysis step are broadcast from rank 0 (RCCL when the tensors live in HBM,
gloo in the CPU tests) and there is no cross-tile dependence.

Nothing here touches the reference's single-process API: a rank simply calls gridpp_amd.optimal_interpolation on
its own tile Grid."""
import numpy as np


def row_tile(ny, rank, world):
    """Rows [row0, row1) of a ny-row grid owned by `rank` (contiguous, balanced to within one row)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return ny * rank // world, ny * (rank + 1) // world


def all_tiles(ny, world):
    return [row_tile(ny, r, world) for r in range(world)]


def weighted_row_tiles(row_weights, world, min_rows=1):
    """Row tiles of unequal height for a field whose rows do not cost the same: [(row0, row1)] * world, contiguous, such that every tile
    carries about 1 / world of sum(row_weights) (each boundary at the row where the running sum crosses its share; at least
    `min_rows` rows per tile while there are enough rows).  Every rank computes the same boundaries from the same weights.
    A cost model for optimal interpolation: row_cost_from_observations()."""
    w = np.maximum(np.asarray(row_weights, dtype=np.float64).ravel(), 0.0)
    ny = w.size
    if world < 1:
        raise ValueError("world must be >= 1")
    if ny == 0:
        return [(0, 0)] * world
    if not np.isfinite(w).all() or w.sum() <= 0:
        return all_tiles(ny, world)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    bounds = [0]
    for r in range(1, world):
        b = int(np.searchsorted(cum, cum[-1] * r / world, side="left"))
        b = max(b, bounds[-1] + min_rows)                       # not before the previous boundary (+ the minimum height) ...
        b = min(b, ny - (world - r) * min_rows)                 # ... and leave the minimum for the tiles that follow
        bounds.append(max(bounds[-1], min(b, ny)))
    bounds.append(ny)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def row_cost_from_observations(row_lats, obs_lats, radius_deg, base=1.0, per_obs=0.0):
    """A simple cost model of an OI row for weighted_row_tiles(): base + per_obs x (observations within `radius_deg` of the row's
    latitude) -- the candidate scan of a tile grows with the observations in range, the solve does not (max_points caps it).
    With per_obs = 0 every row costs the same (equal tiles)."""
    row_lats = np.asarray(row_lats, dtype=np.float64).ravel()
    o = np.sort(np.asarray(obs_lats, dtype=np.float64).ravel())
    n = np.searchsorted(o, row_lats + radius_deg, side="right") - np.searchsorted(o, row_lats - radius_deg, side="left")
    return base + per_obs * n


def broadcast_observations(values, src=0, group=None):
    """In-place broadcast of the packed per-step observation block (e.g. a (3, S) tensor holding obs, ratios,
    background-at-points) from `src`.  `values` is a torch tensor on every rank (CUDA -> RCCL, CPU -> gloo)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(values, src=src, group=group)
    return values


class ObservationStream:
    """Per-step observation blocks from rank 0, buffered ahead: while step k computes from slot k % n, the broadcast of
    step k+1's block into the next slot is already in flight (RCCL's own stream on a GPU box; gloo's worker thread in
    the CPU tests), so the exchange is off the critical path of the kernels.

    slots: n >= 2 equally shaped torch tensors on every rank.  Two slots serve a loop that finishes step k before it asks for
    step k+1; a loop that leaves `ahead` analyses in flight (AnalysisPipeline, gridpp_amd.optimal_interpolation_async) needs
    ahead + 2 slots: block k+1 is posted while the calls of steps k-ahead .. k-1 may not have read their blocks yet.
    fill(slot, step) is called on rank 0 only, right before the block of `step` is posted, and writes that step's values into
    slots[slot] (None = the slots already hold them).  next() returns the tensor holding the values of the next step, valid
    until n - 1 further next() calls."""

    def __init__(self, slots, rank, fill=None, src=0, group=None):
        import torch.distributed as dist
        if len(slots) < 2:
            raise ValueError("ObservationStream needs at least two slots")
        self.slots, self.rank, self.fill, self.src, self.group = slots, rank, fill, src, group
        self.on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.dist = dist
        self.pending = [None] * len(slots)
        self.k = 0

    def _post(self, slot, step):
        if self.rank == self.src and self.fill is not None:
            self.fill(slot, step)
        if self.on:
            self.pending[slot] = self.dist.broadcast(self.slots[slot], src=self.src, group=self.group, async_op=True)

    def next(self):
        k, n = self.k, len(self.slots)
        self.k = k + 1
        cur, nxt = k % n, (k + 1) % n
        if self.on:
            if self.pending[cur] is None:
                self._post(cur, k)
            self.pending[cur].wait()                      # this step's values have arrived
            self.pending[cur] = None
            self._post(nxt, k + 1)                        # the next step's values travel while this step computes
        elif self.fill is not None:
            self.fill(cur, k)
        return self.slots[cur]

    def drain(self):
        """Waits for the one block posted ahead of the last step (call before the final barrier)."""
        for i, w in enumerate(self.pending):
            if w is not None:
                w.wait()
                self.pending[i] = None


class AnalysisPipeline:
    """Keeps deferred analyses in flight (objects with a wait() method: gridpp_amd.optimal_interpolation_async on a GPU box).
    push(pending) hands over the analysis just enqueued and returns the result of the one that has to complete now -- the one pushed
    `ahead` calls earlier -- or None while the pipeline fills; drain() completes the rest, in order.  ahead = 0 is the synchronous loop;
    ahead = 1 keeps the GPU busy with call k while the host waits for call k-1 and prepares call k+1 (an ObservationStream then needs
    three slots: see there).  Every analysis is independent of the others (src/api/oi.cpp:221-338 keeps no state between calls), so the
    results are those of the one-at-a-time loop bit for bit; what the pipeline removes is the idle time of the GPU between two calls (the
    host's read-back, wake-up and launch latencies)."""

    def __init__(self, ahead=1):
        if ahead < 0:
            raise ValueError("ahead must be >= 0")
        self.ahead, self.q = ahead, []

    def push(self, pending):
        self.q.append(pending)
        return self.q.pop(0).wait() if len(self.q) > self.ahead else None

    def drain(self):
        out = [p.wait() for p in self.q]
        self.q = []
        return out


def tiled_optimal_interpolation(lats, lons, background, plats, plons, values, structure_args, max_points,
                                rank, world, compute, allow_extrapolation=True):
    """Runs this rank's row tile of an optimal_interpolation call.

    lats/lons/background: the FULL (Y, X) arrays (every rank can generate or map them; only the tile is touched)
    values: (3, S) tensor [obs, ratios, background_at_points]; only rank 0's content matters, it is broadcast
    compute(lats_tile, lons_tile, bg_tile, plats, plons, obs, ratios, pbg, structure_args, max_points, allow) -> tile
        the single-GPU entry point (gridpp_amd on a GPU box; the CPU tests inject the oracle here)
    returns (row0, row1, analysis_tile)
    """
    row0, row1 = row_tile(np.shape(lats)[0], rank, world)
    values = broadcast_observations(values)
    v = values.detach().cpu().numpy() if hasattr(values, "detach") else np.asarray(values)
    tile = compute(lats[row0:row1], lons[row0:row1], background[row0:row1], plats, plons, v[0], v[1], v[2],
                   structure_args, max_points, allow_extrapolation)
    return row0, row1, tile


def gpu_compute(lats, lons, bg, plats, plons, obs, ratios, pbg, structure_args, max_points, allow_extrapolation=True):
    """The real single-GPU stage: gridpp_amd on the current device."""
    import gridpp_amd as gridpp
    grid = gridpp.Grid(lats, lons)
    points = gridpp.Points(plats, plons)
    return gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, gridpp.BarnesStructure(*structure_args),
                                        max_points, allow_extrapolation)


def halo_rows(ny, rank, world, halfwidth):
    """Rows [lo, hi) a rank must hold to compute the neighbourhood filter on its tile [row0, row1): the tile plus a
    read-only halo of `halfwidth` rows on each side, clipped at the true domain edge (src/api/neighbourhood.cpp:104-107
    clips windows only there).  Returns (lo, hi, row0, row1)."""
    row0, row1 = row_tile(ny, rank, world)
    return max(0, row0 - halfwidth), min(ny, row1 + halfwidth), row0, row1


def tiled_neighbourhood(field, halfwidth, statistic, rank, world, compute):
    """This rank's row tile of neighbourhood(field, halfwidth, statistic): compute on tile + halo, keep the tile.

    compute(sub_field, halfwidth, statistic) -> 2-D array is the single-GPU entry point
    (gridpp_amd.neighbourhood on a GPU box; the CPU tests inject the oracle).  Exact for every statistic: a window
    never reaches beyond `halfwidth` rows, and the clipped windows at the domain edge see the same rows as in the
    single-process call.  (The separable box sums make this possible; the reference's global summed-area table has a
    prefix dependence over the whole image.)"""
    lo, hi, row0, row1 = halo_rows(np.shape(field)[0], rank, world, halfwidth)
    sub = compute(field[lo:hi], halfwidth, statistic)
    return row0, row1, sub[row0 - lo:row0 - lo + (row1 - row0)]


class HaloExchange:
    """Device-side halo of a row-tiled field for the neighbourhood filters: every step each rank sends its first
    and last `halfwidth` rows to the neighbouring ranks and receives theirs -- point-to-point over xGMI (torch.distributed
    batch_isend_irecv = ncclSend / ncclRecv on the nccl = RCCL backend; gloo in the CPU tests), 2 x halfwidth x X x E x 4 bytes
    per interior boundary and no collective.  The padded tile lives in one preallocated buffer: [top halo | tile | bottom halo],
    the tile rows are a view of it, so "exchange" moves only the halo rows.

    tile: (rows, X[, E]) tensor of this rank's rows; exchange() -> (padded, top) with padded[top:top + rows] == tile."""

    def __init__(self, tile, halfwidth, rank, world, group=None):
        import torch
        self.rank, self.world, self.hw, self.group = rank, world, int(halfwidth), group
        rows = tile.shape[0]
        if world > 1 and rows < self.hw:
            raise ValueError("a row tile must hold at least `halfwidth` rows")
        self.top = self.hw if rank > 0 else 0
        self.bot = self.hw if rank < world - 1 else 0
        self.buf = torch.empty((self.top + rows + self.bot,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
        self.tile = self.buf[self.top:self.top + rows]
        self.tile.copy_(tile)

    def exchange(self):
        import torch.distributed as dist
        hw = self.hw
        rows = self.tile.shape[0]
        ops = []
        if self.top:     # neighbour above: my first hw rows go up, its last hw rows come down
            ops.append(dist.P2POp(dist.isend, self.tile[:hw].contiguous(), self.rank - 1, self.group))
            ops.append(dist.P2POp(dist.irecv, self.buf[:hw], self.rank - 1, self.group))
        if self.bot:
            ops.append(dist.P2POp(dist.isend, self.tile[rows - hw:].contiguous(), self.rank + 1, self.group))
            ops.append(dist.P2POp(dist.irecv, self.buf[self.top + rows:], self.rank + 1, self.group))
        if ops and self.buf.is_cuda and dist.get_backend(self.group) == "gloo":
            # gloo has no device point-to-point: only the one-GPU logic test (bench.py with GPP_BENCH_BACKEND=gloo) comes here
            cpu_ops, back = [], []
            for o in ops:
                t = o.tensor.cpu()
                cpu_ops.append(dist.P2POp(o.op, t, o.peer, self.group))
                if o.op is dist.irecv:
                    back.append((o.tensor, t))
            for w in dist.batch_isend_irecv(cpu_ops):
                w.wait()
            for dst, src in back:
                dst.copy_(src)
        elif ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return self.buf, self.top
