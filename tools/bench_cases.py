"""Synthetic workloads of BASELINE.json's configs (shapes, distributions, seeds) and one timed case per config.
Shared by bench.py (the `other_configs` object of the bench line) and tools/bench_paths.py (one JSON line per case).
Every case returns a dict with the same per-step fields: ms, kernel_ms (HIP events where the library reports them), Mcells/s,
algorithmic GB/s and its fraction of the 8 TB/s HBM peak."""
import time

import numpy as np

HBM_PEAK = 8.0e12      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK = 78.6e12    # vector fp64 (the matrix fp64 peak of MI355X is the same figure)


def make_workload(ny, nx, S, seed, row0, row1):
    """[0,1] deg^2 geodetic grid, uniform obs, obs = pbg + N(0,1), ratios U(0.1,1)."""
    rng = np.random.default_rng(seed)
    lat1 = np.linspace(0, 1, ny, dtype=np.float64)[row0:row1]
    lon1 = np.linspace(0, 1, nx, dtype=np.float64)
    lats, lons = np.meshgrid(lat1, lon1, indexing="ij")
    plat, plon = rng.random(S), rng.random(S)
    pbg = rng.normal(0, 1, S).astype(np.float32)
    obs = (pbg + rng.normal(0, 1, S)).astype(np.float32)
    ratios = rng.uniform(0.1, 1, S).astype(np.float32)
    # smooth deterministic background field (any fixed smooth function)
    bg = (np.sin(6 * lats) * np.cos(4 * lons) * 3).astype(np.float32)
    return lats, lons, bg, plat, plon, obs, ratios, pbg


def timeit(fn, reps=3, warm=1):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def terrain(lat, lon):
    """smooth synthetic topography (m) and land area fraction: a few hundred metres of relief over tens of km"""
    z = 500 + 300 * np.sin(lat * 9.0) * np.cos(lon * 7.0) + 150 * np.sin(lat * 31.0 + 1.0) * np.sin(lon * 23.0) + 50 * np.cos(lat * 90.0) * np.cos(lon * 70.0)
    laf = np.clip(0.5 + 0.6 * np.sin(lat * 5.0 + lon * 3.0), 0, 1)
    return z, laf


def oi_inputs(ny, nx, S, seed, elev=False):
    """the OI workload with its terrain variant: (lats, lons, bg, plat, plon, obs, ratios, pbg, ge, gl, pe, pl, v, w)"""
    lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(ny, nx, S, seed, 0, ny)
    rng = np.random.default_rng(seed + 7)
    ge = gl = pe = pl = ()
    v = w = 0
    if elev == "noise":     # white-noise elevation / laf per cell: no two cells of a tile select the same observations
        ge, gl = rng.uniform(0, 1000, (ny, nx)), rng.uniform(0, 1, (ny, nx))
        pe, pl = rng.uniform(0, 1000, S), rng.uniform(0, 1, S)
        v, w = 200, 0.5
    elif elev:              # smooth terrain
        ge, gl = terrain(np.deg2rad(lats) * 40, np.deg2rad(lons) * 40)
        pe, pl = terrain(np.deg2rad(plat) * 40, np.deg2rad(plon) * 40)
        pe = pe + rng.normal(0, 30, S)   # stations are not exactly on the model terrain
        v, w = 200, 0.5
    return lats, lons, bg, plat, plon, obs, ratios, pbg, ge, gl, pe, pl, v, w


def oi_case(name, ny, nx, S, mp, seed, elev=False, reps=3):
    import torch
    import gridpp_amd as gridpp
    lats, lons, bg, plat, plon, obs, ratios, pbg, ge, gl, pe, pl, v, w = oi_inputs(ny, nx, S, seed, elev)
    grid = gridpp.Grid(lats, lons, ge, gl)
    points = gridpp.Points(plat, plon, pe, pl)
    st = gridpp.BarnesStructure(10000, v, w)
    d = [torch.from_numpy(a).cuda() for a in (bg, obs, ratios, pbg)]
    t = timeit(lambda: gridpp.optimal_interpolation(grid, d[0], points, d[1], d[2], d[3], st, mp), reps=reps)
    s = gridpp.oi_last_stats()
    # the same call as a stream of analyses with one ahead (GPP_ASYNC + gpp_wait, what the headline runs): `ms` stays the blocking call
    import time as _time
    from gridpp_amd.dist import AnalysisPipeline
    pipe, K = AnalysisPipeline(1), max(4, 2 * reps)
    for _ in range(2):
        pipe.push(gridpp.optimal_interpolation_async(grid, d[0], points, d[1], d[2], d[3], st, mp))
    pipe.drain()
    torch.cuda.synchronize(); t0 = _time.perf_counter()
    for _ in range(K):
        pipe.push(gridpp.optimal_interpolation_async(grid, d[0], points, d[1], d[2], d[3], st, mp))
    pipe.drain()
    torch.cuda.synchronize(); ta = (_time.perf_counter() - t0) / K
    gbs = ny * nx * 28 / (s["kernel_ms"] * 1e-3) / 1e9   # x, y, z, elev, laf, background read + analysis written
    return {"case": name, "cells": ny * nx, "ms": t * 1e3, "ms_one_analysis_ahead": ta * 1e3, "kernel_ms": s["kernel_ms"], "Mcells/s": ny * nx / t / 1e6,
            "solves": s["solves"], "declined_tiles": s["fallback_tiles"], "items_left_to_k_oi": s["fallback_subtiles"],
            "GB/s_algorithmic": gbs, "frac_hbm": gbs * 1e9 / HBM_PEAK, "bytes_per_cell": 28}


def c4_cube(ny, nx, E):
    import torch
    g = torch.Generator(device="cuda").manual_seed(1003)
    return torch.rand((ny, nx, E), generator=g, device="cuda") * 10


def nb_cases(ny, nx, E, hw, cube=None, qs=(0.5,), two_d=True):
    import torch
    import gridpp_amd as gridpp
    if cube is None:
        cube = c4_cube(ny, nx, E)
    out = []
    bytes_alg = ny * nx * (4 * E + 4)
    t = timeit(lambda: gridpp.neighbourhood(cube, hw, gridpp.Mean))
    out.append({"case": "C4 neighbourhood Mean %dx%dx%d hw=%d" % (ny, nx, E, hw), "cells": ny * nx, "ms": t * 1e3, "Mcells/s": ny * nx / t / 1e6,
                "GB/s_algorithmic": bytes_alg / t / 1e9, "frac_hbm": bytes_alg / t / HBM_PEAK, "bytes_per_cell": 4 * E + 4})
    thr = torch.linspace(0, 10, 11, device="cuda")
    for q in qs:
        t = timeit(lambda: gridpp.neighbourhood_quantile_fast(cube, q, hw, thr))
        out.append({"case": "C4 quantile_fast q=%g T=11 %dx%dx%d hw=%d" % (q, ny, nx, E, hw), "cells": ny * nx, "ms": t * 1e3,
                    "Mcells/s": ny * nx / t / 1e6, "GB/s_algorithmic": bytes_alg / t / 1e9, "frac_hbm": bytes_alg / t / HBM_PEAK,
                    "bytes_per_cell": 4 * E + 4})
    if two_d:
        f2 = cube[:, :, 0].contiguous()
        for stat, nm in ((gridpp.Mean, "Mean"), (gridpp.Max, "Max")):
            t = timeit(lambda: gridpp.neighbourhood(f2, 7, stat))
            out.append({"case": "neighbourhood 2-D %s %dx%d hw=7" % (nm, ny, nx), "cells": ny * nx, "ms": t * 1e3, "Mcells/s": ny * nx / t / 1e6,
                        "GB/s_algorithmic": ny * nx * 8 / t / 1e9, "frac_hbm": ny * nx * 8 / t / HBM_PEAK, "bytes_per_cell": 8})
    return out


def ensi_inputs(ny, nx, E, S, row0=0, row1=None):
    """smooth field + N(0,1) per member, sigma = 1, h = 10 km."""
    import torch
    row1 = ny if row1 is None else row1
    rng = np.random.default_rng(1004)
    lat1 = np.linspace(0, 1, ny)[row0:row1]
    lats, lons = np.meshgrid(lat1, np.linspace(0, 1, nx), indexing="ij")
    base = torch.from_numpy((np.sin(6 * lats) * np.cos(4 * lons) * 3).astype(np.float32)).cuda()
    g = torch.Generator(device="cuda").manual_seed(1004 + row0)
    bg = base[:, :, None] + torch.randn((row1 - row0, nx, E), generator=g, device="cuda")
    plat, plon = rng.random(S), rng.random(S)
    pbg = torch.from_numpy(rng.normal(0, 1, (S, E)).astype(np.float32)).cuda()
    obs = torch.from_numpy(rng.normal(0, 1, S).astype(np.float32)).cuda()
    sig = torch.ones(S, device="cuda")
    return lats, lons, bg, plat, plon, pbg, obs, sig


def ensi_case(ny, nx, E, S, mp, reps=2, converged=True):
    import gridpp_amd as gridpp
    lats, lons, bg, plat, plon, pbg, obs, sig = ensi_inputs(ny, nx, E, S)
    grid = gridpp.Grid(lats, lons)
    points = gridpp.Points(plat, plon)
    st = gridpp.BarnesStructure(10000)
    t = timeit(lambda: gridpp.optimal_interpolation_ensi(grid, bg, points, obs, sig, pbg, st, mp), reps=reps, warm=1)
    kms = gridpp.ensi_last_kernel_ms()
    # the same call with the per-cell Jacobi sweeps run to convergence (gpp_ensi_set_convergence(1)): the mode that meets north_star's
    # plain 1e-5 measure everywhere; the default (`ms`) stops the sweeps at |E| <= 0.040 c and adds a perturbation series of six float32 products (round 4; seven until round 6: no value of
    # the randomised soak outside the plain measure either, worst 2.5e-6 as with converged sweeps)
    tc = float("nan")
    if converged:     # (the profiling tools pass False: their kernel statistics are those of the default mode alone)
        gridpp.ensi_set_convergence(True)
        try:
            tc = timeit(lambda: gridpp.optimal_interpolation_ensi(grid, bg, points, obs, sig, pbg, st, mp), reps=1, warm=0)
        finally:
            gridpp.ensi_set_convergence(False)
    res = {"case": "C5 EnSI %dx%dx%d, %d obs, max_points=%d" % (ny, nx, E, S, mp), "cells": ny * nx, "ms": t * 1e3, "kernel_ms": kms,
           "ms_converged": tc * 1e3, "Mcells/s_converged": ny * nx / tc / 1e6,
           "mode": "ms / Mcells/s: default mode (Jacobi sweeps stopped at |E| <= 0.040 c + six-product float32 perturbation series; plain 1e-5 measure holds on the soak, "
                   "worst 2.5e-6 as with converged sweeps); ms_converged: gpp_ensi_set_convergence(1), sweeps to convergence (plain 1e-5 measure)",
           "Mcells/s": ny * nx / t / 1e6, "GB/s_algorithmic": ny * nx * (8 * E + 16) / t / 1e9, "frac_hbm": ny * nx * (8 * E + 16) / t / HBM_PEAK,
           "bytes_per_cell": 8 * E + 16}
    res.update(ensi_fp64(ny, nx, E, S, mp, kms))
    return res


def ensi_fp64(ny, nx, E, S, mp, kms):
    """FP64 work of the EnSI call: EXECUTED flops per call from the committed rocprofv3 PMC bundle of this workload
    (profiles/hbm_traffic.json: 64 lanes x (ADD_F64 + MUL_F64 + 2 FMA_F64) wave-instructions + 512 x MFMA_MOPS_F64) over the live
    kernel time, against the 78.6 TFLOP/s vector FP64 peak; the reference's own E x E operation count is kept
    beside it as a labelled extra -- the kernels solve an n x n problem (n <= 32) instead, so it is not what they execute."""
    import json, os
    n = mp
    ref_flops = ny * nx * (2.0 * n * E * E + 10.0 * E ** 3 + 2.0 * E * E * n)
    out = {"reference_formulation_TFLOP_per_call": ref_flops / 1e12,
           "reference_formulation_TFLOPs_equivalent": ref_flops / (kms * 1e-3) / 1e12}
    try:
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "hbm_traffic.json")) as f:
            p = json.load(f)["ensi_C5"]
        if p["workload"] == "optimal_interpolation_ensi %dx%dx%d, %d obs, max_points=%d" % (ny, nx, E, S, mp):
            ex = 64.0 * (p["SQ_INSTS_VALU_ADD_F64"] + p["SQ_INSTS_VALU_MUL_F64"] + 2.0 * p["SQ_INSTS_VALU_FMA_F64"]) + 512.0 * p["SQ_INSTS_VALU_MFMA_MOPS_F64"]
            if "SQ_INSTS_VALU_MFMA_MOPS_F32" in p:   # the perturbation series (round 4): float32 products beside the FP64 work, not part of the FP64 fraction
                out["fp32_matrix_TFLOP_executed_per_call"] = 512.0 * p["SQ_INSTS_VALU_MFMA_MOPS_F32"] / 1e12
            out.update({"fp64_TFLOP_executed_per_call": ex / 1e12, "fp64_TFLOPs_executed": ex / (kms * 1e-3) / 1e12,
                        "frac_fp64_peak_executed": ex / (kms * 1e-3) / FP64_PEAK, "fp64_source": "committed_profile " + p.get("_source", "")})
    except (OSError, KeyError, ValueError):
        pass
    return out


# ---- CPU baselines of C4 / C5 (bench.py: `cpu_baseline` inside their other_configs entries) -------------------------------------------------
def _usable_threads():
    import os
    from oracle import oracle as O
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q = f.read().split()
        if q[0] != "max":
            n = max(1, min(n, int(float(q[0]) / float(q[1]))))
    except (OSError, ValueError, IndexError):
        pass
    return max(1, min(n, O.omp_max_threads()))


def c4_cpu_baseline(nx, E, hw, target_s=8.0):
    """The CPU oracle on a band of rows of the C4 cube (same distribution: U(0, 10) members), with OpenMP exactly where the reference has it
    (neighbourhood.cpp:101 window loop; :453 one thread per threshold, :483 rows): the member statistic and the summed-area table are
    serial in the reference, so are they here.  One thread and all usable threads."""
    from oracle import oracle as O
    rng = np.random.default_rng(1003)
    thr = np.linspace(0, 10, 11).astype(np.float32)
    threads = _usable_threads()
    out = {}
    band = {}
    for name, fn in (("mean", lambda c: O.neighbourhood(c, hw, O.Mean)), ("quantile_fast", lambda c: O.neighbourhood_quantile_fast(c, 0.5, hw, thr))):
        rows = 4 * hw + 2
        cube = (rng.random((rows, nx, E), dtype=np.float32) * 10)
        t0 = time.perf_counter(); fn(cube); per_row = (time.perf_counter() - t0) / rows
        rows = int(max(4 * hw + 2, min(320, 0.3 * target_s / per_row)))   # (<= 0.5 GB of members: generating them is not the point)
        cube = (rng.random((rows, nx, E), dtype=np.float32) * 10)

        def timed(budget):      # the band again and again until `budget` seconds are spent: seconds per pass
            n, t0 = 0, time.perf_counter()
            while True:
                fn(cube); n += 1
                dt = time.perf_counter() - t0
                if dt >= budget:
                    return dt / n, n
        t1, n1 = timed(0.25 * target_s)
        O.set_neighbourhood_threads(threads)
        try:
            tn, nn = timed(0.25 * target_s)
        finally:
            O.set_neighbourhood_threads(1)
        band[name] = rows
        out[name] = {"value": rows * nx / tn, "unit": "cells/s", "cores": threads, "kind": "port", "one_thread_value": rows * nx / t1,
                     "sample": "a band of %d rows x %d columns x %d members of the same distribution, %d passes on %d threads (%.2f s each), %d passes on 1 thread (%.2f s each)" % (rows, nx, E, nn, threads, tn, n1, t1),
                     "algorithm": "oracle/gridpp_oracle.c: member statistic + summed-area table (serial, as in the reference) + window loop; OpenMP where "
                                  "neighbourhood.cpp:101,453,483 have it"}
    return out


def c5_cpu_baseline(ny, nx, E, S, mp, target_s=10.0):
    """The CPU oracle's optimal_interpolation_ensi on a sample of grid points of the C5 workload.  The reference runs this loop SERIALLY
    (oi_ensi.cpp:204-207: its `#pragma omp parallel for` is commented out), so `value` is the one-thread figure; `all_threads_value` runs
    disjoint ranges of the sample on all usable threads (what the pragma would give)."""
    import threading
    from oracle import oracle as O
    rng = np.random.default_rng(1004)
    plat, plon = rng.random(S), rng.random(S)
    pbg = rng.normal(0, 1, (S, E)).astype(np.float32)
    obs = rng.normal(0, 1, S).astype(np.float32)
    sig = np.ones(S, np.float32)
    op, st = O.Pts(plat, plon), O.Barnes(10000.0)
    threads = _usable_threads()

    def sample(n, seed):
        r = np.random.default_rng(seed)
        la, lo = r.random(n), r.random(n)
        bg = ((np.sin(6 * la) * np.cos(4 * lo) * 3)[:, None] + r.normal(0, 1, (n, E))).astype(np.float32)
        return O.Pts(la, lo), bg
    g, bg = sample(64, 1)
    t0 = time.perf_counter(); O.oi_ensi(g, bg, op, obs, sig, pbg, st, mp); per_cell = (time.perf_counter() - t0) / 64
    n1 = int(max(64, min(20000, 0.35 * target_s / per_cell)))
    g, bg = sample(n1, 2)
    t0 = time.perf_counter(); O.oi_ensi(g, bg, op, obs, sig, pbg, st, mp); t1 = time.perf_counter() - t0
    nall = n1 * max(1, min(threads, 8))
    parts = [sample(nall // threads + 1, 10 + k) for k in range(threads)]
    ws = [threading.Thread(target=lambda gb=gb: O.oi_ensi(gb[0], gb[1], op, obs, sig, pbg, st, mp)) for gb in parts]
    t0 = time.perf_counter()
    for w in ws: w.start()
    for w in ws: w.join()
    ta = time.perf_counter() - t0
    ncells = sum(gb[1].shape[0] for gb in parts)
    return {"value": n1 / t1, "unit": "cells/s", "cores": 1, "kind": "port",
            "all_threads_value": ncells / ta, "all_threads_cores": threads,
            "sample": "%d random grid points of the same workload (%d members, %d obs, max_points %d) on 1 thread in %.1f s; %d points over %d threads in %.1f s"
                      % (n1, E, S, mp, t1, ncells, threads, ta),
            "algorithm": "oracle/gridpp_oracle.c:orc_oi_ensi_core (E x E inverse + cyclic-Jacobi eig_sym per grid point); the reference's loop is serial "
                         "(oi_ensi.cpp:204-207)"}
