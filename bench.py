#!/usr/bin/env python
"""bench.py -- throughput of the optimal-interpolation hot path on MI355X.

Metric (BASELINE.json): grid cells / s of gridpp.optimal_interpolation on a 4000x4000 grid with
10 000 observations, BarnesStructure(10000), max_points = 30 (configs[2]; it fits one GPU, so it is
also the N=1 workload).  A "step" is one full optimal_interpolation pass over the grid.  With N > 1
the grid is row-tiled over the ranks (strong scaling on the fixed grid), the per-step observation
values are broadcast from rank 0 over RCCL, and there is no other data-path collective.

Inputs are synthetic (seed 1002) and resident in HBM (torch CUDA tensors handed to the
C-ABI as device pointers) when the timed region starts.

The ONE JSON line of rank 0 carries, besides the contract keys:
  roofline          HBM view of the dominant kernel (algorithmic 28 B/cell over its HIP-event duration)
  roofline_compute  the bound that actually binds OI: VALU issue (FP64 and other VALU wave-instructions per launch from the
                    committed rocprofv3 PMC bundle, 4 and 2 issue cycles each, over the live kernel time and 1024 SIMDs)
  host_inclusive    the same call from numpy buffers (PCIe both ways; never `value`)
  other_configs     the other BASELINE.json configs with the same per-step fields (N = 1 only)
  cpu_baseline      the oracle's OpenMP / cell-list build of the same loop on this box's host cores (1 thread and all threads)

`--case ensi` / `--case nbh` run config 5 / config 4 instead of OI with the same contract (row tiles; EnSI broadcasts the
observation block, the neighbourhood filter exchanges halo rows between neighbouring ranks with RCCL send / recv).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from tools.bench_cases import make_workload, HBM_PEAK, FP64_PEAK  # noqa: E402

BYTES_PER_CELL = 28   # SURVEY.md 8(d), the x/y/z form: the kernel reads x, y, z, elev, laf, background (6 x 4 B) and writes the analysis (4 B)
HBM_PEAK_GBS = HBM_PEAK / 1e9
# MI355X_MICROARCH.md "Wave scheduling": 256 CUs x 4 SIMD-32 at 2.4 GHz; a wave64 VALU instruction issues over 2 cycles,
# an FP64 one over 4 (vector FP64 peak = half the FP32 peak)
SIMDS, CLOCK_HZ, CYC_VALU, CYC_FP64 = 1024, 2.4e9, 2.0, 4.0


def host_cpu_info():
    """what this process may use of the host: affinity mask and cgroup quota (os.cpu_count() is the machine, not the lease)"""
    info = {"os_cpu_count": os.cpu_count()}
    try:
        info["sched_getaffinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        info["sched_getaffinity"] = None
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                quota = None if txt[0] == "max" else float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                    quota = None if q < 0 else q / float(f2.read().split()[0])
            info["cgroup_cpu_quota_cores"] = quota
            info["cgroup_source"] = path
            break
        except (OSError, ValueError, IndexError):
            continue
    return info


def cpu_baseline(ny, nx, S, seed, h, max_points, target_s=12.0):
    """The CPU oracle's OpenMP build of the reference loop (oracle/gridpp_oracle.c: orc_oi_full_omp -- cell-list radius query,
    dynamic schedule over the grid points; bit-identical to the serial oracle, tests/test_oracle_baseline.py) timed on this
    box's host cores on a bounded sample of the same workload: every k-th grid row."""
    from oracle import oracle as O
    lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(ny, nx, S, seed, 0, ny)
    hw = host_cpu_info()
    usable = hw.get("sched_getaffinity") or os.cpu_count() or 1
    if hw.get("cgroup_cpu_quota_cores"):
        usable = max(1, min(usable, int(hw["cgroup_cpu_quota_cores"])))
    threads = max(1, min(usable, O.omp_max_threads()))
    op = O.Pts(plat, plon)
    st = O.Barnes(h)

    def run(nrows, thr):
        rows = np.unique(np.linspace(0, ny - 1, nrows).astype(int))
        g = O.Pts(lats[rows].ravel(), lons[rows].ravel())
        b = bg[rows].ravel()
        t0 = time.perf_counter()
        O.oi_baseline(g, b, op, obs, ratios, pbg, st, max_points, threads=thr)
        return rows.size * nx, time.perf_counter() - t0

    # one thread: a few rows, sized from a one-row calibration; all threads: sized from a short all-thread calibration (the
    # scaling over the threads of a shared box is not known in advance)
    c0, t0 = run(1, 1)
    per_cell = t0 / c0
    n1 = max(1, min(ny, int(0.25 * target_s / per_cell / nx)))
    c1, t1 = run(n1, 1)
    one = c1 / t1
    cc, tc = run(max(2, threads // 8), threads)
    nall = max(2, min(ny, int(0.6 * target_s * (cc / tc) / nx)))
    ca, ta = run(nall, threads)
    return {"value": ca / ta, "unit": "cells/s", "cores": threads, "kind": "port",
            "one_thread_value": one, "host": hw, "scaling_over_threads": (ca / ta) / one,
            "sample": "%d of %d grid rows (%d cells) of the same workload on %d threads in %.1f s; 1 thread: %d rows in %.1f s"
                      % (ca // nx, ny, ca, threads, ta, c1 // nx, t1),
            "algorithm": "cell-list radius query + per-point double inverse, OpenMP dynamic schedule (oracle/gridpp_oracle.c:orc_oi_full_omp)"}


def pmc_profile(workload, world):
    """Counters of the dominant kernel per launch from the committed rocprofv3 PMC passes of this exact workload (profiles/)."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            t = json.load(f)["k_oi_union"]
        if t["workload"] == workload and t["n_gpus"] == world:
            return t
    except (OSError, KeyError, ValueError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--case", choices=["oi", "ensi", "nbh"], default="oi")
    ap.add_argument("--ny", type=int, default=0)
    ap.add_argument("--nx", type=int, default=0)
    ap.add_argument("--obs", type=int, default=0)
    ap.add_argument("--max-points", type=int, default=30)
    ap.add_argument("--h", type=float, default=10000.0)
    ap.add_argument("--equal-tiles", action="store_true", help="case oi, N > 1: keep the equal row tiles (no rebalancing by measured kernel time)")
    ap.add_argument("--sync-calls", action="store_true", help="case oi: one blocking call per step instead of one analysis ahead (A/B of the GPP_ASYNC path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import gridpp_amd as gridpp
    from gridpp_amd import dist as gdist

    # GPP_* variables select implementations inside the library (tests, A/B timing): a benchmark line must not carry any.
    # (GPP_BENCH_* are this script's own: the N > 1 logic test on a one-GPU box; they are recorded in the line.)
    lib_overrides = gridpp.active_overrides()
    overrides = lib_overrides + sorted(k for k in os.environ if k.startswith("GPP_BENCH_"))
    if lib_overrides:
        raise SystemExit("bench.py: library overrides are set in the environment: %s -- unset them" % ", ".join(lib_overrides))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    dist = None
    # GPP_BENCH_SHARE_GPU=1 + GPP_BENCH_BACKEND=gloo: every rank on GPU 0, exchange over gloo -- only to exercise the N > 1 rank
    # logic on a one-GPU box (RCCL refuses two ranks on one device); never the measured configuration
    share = os.environ.get("GPP_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    gridpp.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ   # under torch.distributed.run, also for N = 1
    backend = os.environ.get("GPP_BENCH_BACKEND", "nccl")
    if world > 1 or launched:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    case = args.case
    seed = 1002
    extra = {}
    if case == "oi":
        ny, nx, S = args.ny or 4000, args.nx or 4000, args.obs or 10000
        row0, row1 = gdist.row_tile(ny, rank, world)      # contiguous row tile of this rank
        lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(ny, nx, S, seed, row0, row1)
        grid = gridpp.Grid(lats, lons)                 # x/y/z resident in HBM
        points = gridpp.Points(plat, plon)             # bin-sorted observation index resident in HBM
        structure = gridpp.BarnesStructure(args.h)
        d_bg = torch.from_numpy(bg).to(dev)
        tiles_note = None
        if world > 1 and not args.equal_tiles:
            # Row tiles by measured cost (round 5): equal tiles do not cost the same (the ranks at the edge of the domain see fewer candidates per
            # tile than the ones in the middle) and the step ends with the slowest rank.  Five analyses on the equal tiles, the best kernel time of the last three of
            # every rank gathered, and -- if they are more than 3 % apart -- the rows cut again where the running cost crosses each rank's
            # share (gridpp_amd.dist.weighted_row_tiles; every rank computes the same boundaries from the same gathered times).
            t_obs, t_rat, t_pbg = (torch.from_numpy(a).to(dev) for a in (obs, ratios, pbg))
            probe = []
            for _ in range(5):
                gridpp.optimal_interpolation(grid, d_bg, points, t_obs, t_rat, t_pbg, structure, args.max_points)
                probe.append(gridpp.oi_last_stats()["kernel_ms"])
            mine = torch.tensor([min(probe[2:])], dtype=torch.float64, device=dev)     # (the best of three warm calls: clocks and caches settled)
            allms = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allms, mine)
            allms = [float(t.item()) for t in allms]
            spread = (max(allms) - min(allms)) / (sum(allms) / world)
            tiles_note = {"equal_tiles_kernel_ms": allms, "spread": spread, "rebalanced": False}
            # (GPP_BENCH_FORCE_REBALANCE=1: the logic test of tests/test_gpu_bench_contract.py -- ranks sharing one GPU are further apart than any workload)
            if 0.03 < spread < 0.5 or (os.environ.get("GPP_BENCH_FORCE_REBALANCE") == "1" and spread > 0):      # (beyond that it is not the workload -- ranks sharing a device, a disturbed box --: the equal tiles stay)
                w = np.concatenate([np.full(r1 - r0, allms[r] / max(1, r1 - r0)) for r, (r0, r1) in enumerate(gdist.all_tiles(ny, world))])
                tiles = gdist.weighted_row_tiles(w, world, min_rows=8)
                row0, row1 = tiles[rank]
                lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(ny, nx, S, seed, row0, row1)
                grid = gridpp.Grid(lats, lons)
                d_bg = torch.from_numpy(bg).to(dev)
                tiles_note.update(rebalanced=True, row_tiles=[list(t) for t in tiles])
        # rank 0 owns the observation values of each step; the others receive them over RCCL.  Double-buffered: the broadcast
        # of the NEXT step's values is in flight on RCCL's stream while this step's kernels run on the library stream.
        # Construction, not a step: two analyses on the final geometry so that its memory (which tiles the first pass declines: the list passes then run
        # beside the first pass) and the library's call-to-call workspaces exist before the first warm-up step, whatever --warmup is.
        _t = [torch.from_numpy(a).to(dev) for a in (obs, ratios, pbg)]
        gridpp.optimal_interpolation(grid, d_bg, points, _t[0], _t[1], _t[2], structure, args.max_points)
        if not args.sync_calls:   # (the streams, events and page-locked slots of the deferred calls are created by the first of them)
            for _ in range(2):
                gridpp.optimal_interpolation_async(grid, d_bg, points, _t[0], _t[1], _t[2], structure, args.max_points).wait()
        else:
            gridpp.optimal_interpolation(grid, d_bg, points, _t[0], _t[1], _t[2], structure, args.max_points)
        torch.cuda.synchronize()
        del _t
        host_vals = np.stack([obs, ratios, pbg])
        # One analysis ahead (round 5): the call of step k is ENQUEUED (gpp_optimal_interpolation_full with GPP_ASYNC) and the host waits for the
        # call of step k - 1 -- the GPU goes from one analysis to the next without waiting for the host's read-back, wake-up and launch
        # latencies (~50 us of a 0.6 ms step at 500 rows per rank).  All K analyses complete inside the timed region (drain before the fence).
        # Three observation slots: block k + 1 is posted while call k - 1 may not have read its block yet (gridpp_amd/dist.py).
        ahead = 0 if args.sync_calls else 1
        d_vals = [torch.from_numpy(host_vals).to(dev) if rank == 0 else torch.empty((3, S), dtype=torch.float32, device=dev) for _ in range(ahead + 2)]
        stream = gdist.ObservationStream(d_vals, rank)
        pipe = gdist.AnalysisPipeline(ahead)
        kernel_ms, union_ms = [], []
        last_out = [None]

        def completed(pend):
            st_ = pend.stats()
            kernel_ms.append(st_["kernel_ms"]); union_ms.append(st_["union_kernel_ms"])
            last_out[0] = pend.wait()

        class _Tracked:
            def __init__(self, pend): self.pend = pend
            def wait(self): completed(self.pend); return last_out[0]

        def step():
            v = stream.next()
            pipe.push(_Tracked(gridpp.optimal_interpolation_async(grid, d_bg, points, v[0], v[1], v[2], structure, args.max_points)))
            return last_out[0]

        def drain_steps():
            pipe.drain()
            return last_out[0]
        cells_total, cells_rank = ny * nx, (row1 - row0) * nx
        workload = "optimal_interpolation %dx%d grid, %d obs, BarnesStructure(%g), max_points=%d" % (ny, nx, S, args.h, args.max_points)
        metric = "grid cells/sec for optimal_interpolation, 4000x4000 grid, 10k obs"
        dtype = "f32 (rho, distances) + f64 (local solve)"
    elif case == "ensi":
        from tools.bench_cases import ensi_inputs
        ny, nx, S, E = args.ny or 2500, args.nx or 2500, args.obs or 5000, 50
        row0, row1 = gdist.row_tile(ny, rank, world)
        lats, lons, bg, plat, plon, pbg, obs, sig = ensi_inputs(ny, nx, E, S, row0, row1)
        grid = gridpp.Grid(lats, lons)
        points = gridpp.Points(plat, plon)
        structure = gridpp.BarnesStructure(args.h)
        # per-step observation block from rank 0: [obs | sigma | background at the points (S x E)] in one tensor, one broadcast
        block = torch.cat([obs.reshape(S, 1), sig.reshape(S, 1), pbg], dim=1).contiguous()
        slots = [block.clone() if rank == 0 else torch.empty_like(block) for _ in range(2)]
        stream = gdist.ObservationStream(slots, rank)
        kernel_ms, union_ms = [], []

        def step():
            v = stream.next()
            out = gridpp.optimal_interpolation_ensi(grid, bg, points, v[:, 0].contiguous(), v[:, 1].contiguous(), v[:, 2:].contiguous(),
                                                    structure, args.max_points)
            kernel_ms.append(gridpp.ensi_last_kernel_ms()); union_ms.append(kernel_ms[-1])
            return out
        cells_total, cells_rank = ny * nx, (row1 - row0) * nx
        workload = "optimal_interpolation_ensi %dx%d grid x %d members, %d obs, BarnesStructure(%g), max_points=%d" % (ny, nx, E, S, args.h, args.max_points)
        metric = "grid cells/sec for optimal_interpolation_ensi, 2500x2500 grid, 50 members, 5k obs"
        dtype = "f32 (rho, distances, member update accumulation) + f64 (local eigenproblem)"
        extra["bytes_per_cell"] = 8 * E + 16
    else:
        from tools.bench_cases import c4_cube
        ny, nx, E, hw = args.ny or 4000, args.nx or 4000, 100, 15
        row0, row1 = gdist.row_tile(ny, rank, world)
        # every rank generates its own rows of the cube (the data are synthetic); halo rows come from the neighbours each step
        g = torch.Generator(device="cuda").manual_seed(1003 + rank)
        tile = torch.rand((row1 - row0, nx, E), generator=g, device=dev) * 10
        halo = gdist.HaloExchange(tile, hw, rank, world) if world > 1 else None
        stream = None
        kernel_ms, union_ms = [], []

        def step():
            if halo is None:
                return gridpp.neighbourhood(tile, hw, gridpp.Mean)
            padded, top = halo.exchange()                       # ncclSend / ncclRecv of hw rows per boundary
            out = gridpp.neighbourhood(padded, hw, gridpp.Mean)
            return out[top:top + (row1 - row0)]
        cells_total, cells_rank = ny * nx, (row1 - row0) * nx
        workload = "neighbourhood Mean %dx%dx%d, halfwidth %d" % (ny, nx, E, hw)
        metric = "grid cells/sec for neighbourhood(Mean), 4000x4000 grid x 100 members, halfwidth 15"
        dtype = "f32 (member means) + f64 (box sums)"
        extra["bytes_per_cell"] = 4 * E + 4

    finish = drain_steps if case == "oi" else (lambda: None)
    for _ in range(args.warmup):
        step()
    finish()
    kernel_ms.clear(); union_ms.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    if case == "oi":
        out = finish()                                        # the analysis still in flight behind the last step (all K complete inside the timed region)
    if stream is not None:
        stream.drain()                                        # the one broadcast posted ahead of the last step
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(torch.as_tensor(out)).all())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = cells_total * args.steps / dt
        par = {"oi": "row-tiles x%d, obs broadcast over RCCL (double-buffered, overlapped with the kernels)",
               "ensi": "row-tiles x%d, observation block (obs, sigma, S x E background) broadcast over RCCL (double-buffered)",
               "nbh": "row-tiles x%d, halfwidth-row halos exchanged between neighbouring ranks with RCCL send / recv every step"}[case] % world
        res = {
            "metric": metric, "value": value, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": workload, "parallelism": par if world > 1 else "1 GPU",
                       "calls": ("one analysis ahead (GPP_ASYNC + gpp_wait): call k is enqueued while call k - 1 completes" if case == "oi" and not args.sync_calls else "one blocking call per step"),
                       "inputs": "resident in HBM (device pointers through the C-ABI)"},
            "n_ranks_seen": {"env_WORLD_SIZE": world, "torch_distributed": (dist.get_world_size() if dist is not None else 1),
                             "backend": (backend if dist is not None else None)},
            "env_overrides": overrides,
        }
        if case == "oi":
            stats = gridpp.oi_last_stats()
            all_ms = float(np.mean(kernel_ms))            # every kernel of the call (hipEvents on the library stream)
            k_ms = float(np.mean(union_ms))               # the dominant one: k_oi_union, first pass (all tiles)
            k_name = "k_oi_union<true, false, 32>"
            if k_ms <= 0:                                  # that kernel was not used (GPP_OI_NO_UNION): k_oi did everything
                k_ms, k_name = all_ms, "k_oi<32, false, true, false>"
            achieved = cells_rank * BYTES_PER_CELL / (k_ms * 1e-3) / 1e9
            prof = pmc_profile(workload, world)
            res["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                               "traffic": prof.get("traffic_bytes") if prof else None,
                               "note": "OI is instruction-issue-bound by construction (observations stay on-chip): see roofline_compute; algorithmic bytes = 28 B/cell (x, y, z, elev, laf, background read; analysis written); traffic = rocprofv3 FETCH_SIZE*2 + WRITE_SIZE per launch (profiles/)"}
            if prof and "SQ_INSTS_VALU" in prof:
                # issue cycles = FP64 instructions x 4 + the other VALU instructions x 2, against every SIMD issuing every cycle
                valu = prof["SQ_INSTS_VALU"]
                fp64 = prof.get("SQ_INSTS_VALU_FP64")      # ADD_F64 + MUL_F64 + FMA_F64 + TRANS_F64 of the same PMC bundle
                t = k_ms * 1e-3
                if fp64 is None:   # (a bundle without the FP64 counters: the true figure lies between the two bounds)
                    cyc_lo, cyc_hi = valu * CYC_VALU, valu * CYC_FP64
                    frac = cyc_lo / (SIMDS * CLOCK_HZ * t)
                else:
                    cyc_lo = cyc_hi = fp64 * CYC_FP64 + (valu - fp64) * CYC_VALU
                    frac = cyc_lo / (SIMDS * CLOCK_HZ * t)
                res["roofline_compute"] = {"bound": "valu_issue", "unit": "SIMD issue cycles", "achieved": cyc_lo / t / 1e9, "peak": SIMDS * CLOCK_HZ / 1e9,
                                           "achieved_unit": "G issue cycles/s", "frac": frac,
                                           "frac_bounds": [cyc_lo / (SIMDS * CLOCK_HZ * t), cyc_hi / (SIMDS * CLOCK_HZ * t)],
                                           "valu_wave_instructions_per_launch": valu, "fp64_wave_instructions_per_launch": fp64,
                                           "salu_wave_instructions_per_launch": prof.get("SQ_INSTS_SALU"),
                                           "lds_wave_instructions_per_launch": prof.get("SQ_INSTS_LDS"),
                                           "instructions_per_cell": (valu + prof.get("SQ_INSTS_SALU", 0) + prof.get("SQ_INSTS_LDS", 0)) / cells_rank,
                                           "wave_cycle_shares": prof.get("wave_cycle_shares"),
                                           "source": "committed_profile", "profile": prof.get("_source", "profiles/hbm_traffic.json"),
                                           "note": "counters per launch from the committed rocprofv3 PMC bundle of this workload (not re-measured in this run) over the live kernel time; "
                                                   "cycles per wave64 instruction: 2 (FP32 / integer VALU), 4 (FP64), MI355X_MICROARCH.md; frac = sum(instructions x cycles) / (1024 SIMDs x 2.4 GHz x t)"}
            res["kernel"] = {"name": k_name, "avg_ms": k_ms, "all_oi_kernels_ms": all_ms, "cells_per_launch": cells_rank,
                             "factorisations_per_launch": stats["solves"], "cells_updated": stats["cells_updated"],
                             "tiles_declined_by_first_pass": stats["fallback_tiles"], "subtiles_left_to_k_oi": stats["fallback_subtiles"]}
            if tiles_note is not None:
                res["config"]["row_tiles"] = tiles_note
        else:
            bpc = extra["bytes_per_cell"]
            k_ms = float(np.mean(kernel_ms)) if kernel_ms else ms_per_step
            achieved = cells_rank * bpc / (k_ms * 1e-3) / 1e9
            res["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                               "traffic": None, "note": "algorithmic bytes = %d B/cell" % bpc}
            res["kernel"] = {"avg_ms": k_ms, "cells_per_launch": cells_rank}
            if case == "ensi":
                from tools.bench_cases import ensi_fp64
                f64 = ensi_fp64(ny, nx, 50, S, args.max_points, k_ms) if world == 1 else {}
                if "fp64_TFLOPs_executed" in f64:
                    res["roofline_compute"] = {"bound": "fp64", "achieved": f64["fp64_TFLOPs_executed"], "peak": FP64_PEAK / 1e12, "unit": "TFLOP/s",
                                               "frac": f64["frac_fp64_peak_executed"], "source": f64["fp64_source"],
                                               "reference_formulation_TFLOPs_equivalent": f64["reference_formulation_TFLOPs_equivalent"],
                                               "note": "EXECUTED FP64 flops (vector ADD / MUL / FMA + MFMA-F64) per call from the committed PMC bundle over the live kernel time; "
                                                       "the reference's E x E count is kept only as a labelled equivalent"}
        if world == 1 and case == "oi":
            # the same call from numpy buffers (GPP_MEM_HOST): PCIe both ways included -- reported, never `value`
            hl, hlo, hbg, hplat, hplon, hobs, hrat, hpbg = make_workload(ny, nx, S, seed, 0, ny)
            hs = []
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                gridpp.optimal_interpolation(grid, hbg, points, hobs, hrat, hpbg, structure, args.max_points)
                hs.append(time.perf_counter() - t1)
            res["host_inclusive"] = {"ms_per_step": min(hs) * 1e3, "value": cells_total / min(hs), "unit": "cells/s",
                                     "note": "numpy float32 buffers in, numpy out: 64 MB over PCIe each way inside the call"}
        if world == 1 and not args.no_other_configs and case == "oi":
            from tools import bench_cases as bc
            oc = []
            try:
                oc.append(bc.oi_case("C1 OI 200x200, 10 obs, mp=10", 200, 200, 10, 10, 1000))
                oc.append(bc.oi_case("C2 OI 1000x1000, 1k obs, mp=20", 1000, 1000, 1000, 20, 1001))
                oc.append(bc.oi_case("C3 OI 4000x4000, 10k obs, mp=30, smooth terrain elev+laf (v=200,w=0.5)", 4000, 4000, 10000, 30, 1002, elev=True))
                oc.append(bc.oi_case("C3 OI 4000x4000, 10k obs, mp=30, white-noise elev+laf (v=200,w=0.5)", 4000, 4000, 10000, 30, 1002, elev="noise", reps=2))
                oc.extend(bc.nb_cases(4000, 4000, 100, 15, two_d=False))
                torch.cuda.empty_cache()
                oc.append(bc.ensi_case(2500, 2500, 50, 5000, 30))
            except Exception as e:   # a failing secondary case must not take the headline line with it
                oc.append({"case": "error", "message": repr(e)[:300]})
            if not args.no_cpu_baseline:
                # the CPU oracle beside C4 and C5 as well (bounded samples)
                try:
                    c4 = bc.c4_cpu_baseline(4000, 100, 15, target_s=0.5 * args.cpu_seconds)
                    c5 = bc.c5_cpu_baseline(2500, 2500, 50, 5000, 30, target_s=0.7 * args.cpu_seconds)
                    for o in oc:
                        nm = o.get("case", "")
                        base = c4["mean"] if nm.startswith("C4 neighbourhood Mean") else (c4["quantile_fast"] if nm.startswith("C4 quantile_fast") else (c5 if nm.startswith("C5 EnSI") else None))
                        if base is not None:
                            o["cpu_baseline"] = base
                            o["speedup_vs_cpu_baseline"] = o["Mcells/s"] * 1e6 / base["value"]
                except Exception as e:
                    oc.append({"case": "cpu_baseline error", "message": repr(e)[:300]})
            res["other_configs"] = oc
        if world == 1 and not args.no_cpu_baseline and case == "oi":
            res["cpu_baseline"] = cpu_baseline(ny, nx, S, seed, args.h, args.max_points, args.cpu_seconds)
            res["speedup_vs_cpu_baseline"] = value / res["cpu_baseline"]["value"]
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
