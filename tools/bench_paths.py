#!/usr/bin/env python
"""Secondary measurements (not the bench.py contract): the other BASELINE.json configs on one MI355X,
inputs resident in HBM.  Prints one JSON line per case."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gridpp_amd as gridpp  # noqa: E402
from bench import make_workload  # noqa: E402


def timeit(fn, reps=3, warm=1):
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


def oi_case(name, ny, nx, S, mp, seed, elev=False):
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
    grid = gridpp.Grid(lats, lons, ge, gl)
    points = gridpp.Points(plat, plon, pe, pl)
    st = gridpp.BarnesStructure(10000, v, w)
    d = [torch.from_numpy(a).cuda() for a in (bg, obs, ratios, pbg)]
    t = timeit(lambda: gridpp.optimal_interpolation(grid, d[0], points, d[1], d[2], d[3], st, mp))
    s = gridpp.oi_last_stats()
    print(json.dumps({"case": name, "cells": ny * nx, "ms": t * 1e3, "kernel_ms": s["kernel_ms"], "Mcells/s": ny * nx / t / 1e6,
                      "solves": s["solves"], "declined_tiles": s["fallback_tiles"], "items_left_to_k_oi": s["fallback_subtiles"], "GB/s_algorithmic": ny * nx * 24 / (s["kernel_ms"] * 1e-3) / 1e9}), flush=True)


def nb_case(ny, nx, E, hw):
    g = torch.Generator(device="cuda").manual_seed(1003)
    cube = torch.rand((ny, nx, E), generator=g, device="cuda") * 10
    bytes_alg = ny * nx * (4 * E + 4)
    t = timeit(lambda: gridpp.neighbourhood(cube, hw, gridpp.Mean))
    print(json.dumps({"case": "C4 neighbourhood Mean %dx%dx%d hw=%d" % (ny, nx, E, hw), "ms": t * 1e3, "Mcells/s": ny * nx / t / 1e6,
                      "GB/s_algorithmic": bytes_alg / t / 1e9, "frac_hbm_8TBs": bytes_alg / t / 8e12}), flush=True)
    thr = torch.linspace(0, 10, 11, device="cuda")
    for q in (0.5, 0.9):
        t = timeit(lambda: gridpp.neighbourhood_quantile_fast(cube, q, hw, thr))
        print(json.dumps({"case": "C4 quantile_fast q=%g T=11 %dx%dx%d hw=%d" % (q, ny, nx, E, hw), "ms": t * 1e3,
                          "Mcells/s": ny * nx / t / 1e6, "GB/s_algorithmic": bytes_alg / t / 1e9, "frac_hbm_8TBs": bytes_alg / t / 8e12}), flush=True)
    f2 = cube[:, :, 0].contiguous()
    for stat, nm in ((gridpp.Mean, "Mean"), (gridpp.Max, "Max")):
        t = timeit(lambda: gridpp.neighbourhood(f2, 7, stat))
        print(json.dumps({"case": "neighbourhood 2-D %s %dx%d hw=7" % (nm, ny, nx), "ms": t * 1e3, "Mcells/s": ny * nx / t / 1e6,
                          "GB/s_algorithmic": ny * nx * 8 / t / 1e9}), flush=True)


def ensi_case(ny, nx, E, S, mp):
    rng = np.random.default_rng(1004)
    lats, lons = np.meshgrid(np.linspace(0, 1, ny), np.linspace(0, 1, nx), indexing="ij")
    base = torch.from_numpy((np.sin(6 * lats) * np.cos(4 * lons) * 3).astype(np.float32)).cuda()
    g = torch.Generator(device="cuda").manual_seed(1004)
    bg = base[:, :, None] + torch.randn((ny, nx, E), generator=g, device="cuda")
    plat, plon = rng.random(S), rng.random(S)
    pbg = torch.from_numpy(rng.normal(0, 1, (S, E)).astype(np.float32)).cuda()
    obs = torch.from_numpy(rng.normal(0, 1, S).astype(np.float32)).cuda()
    sig = torch.ones(S, device="cuda")
    grid = gridpp.Grid(lats, lons)
    points = gridpp.Points(plat, plon)
    st = gridpp.BarnesStructure(10000)
    t = timeit(lambda: gridpp.optimal_interpolation_ensi(grid, bg, points, obs, sig, pbg, st, mp), reps=2, warm=1)
    print(json.dumps({"case": "C5 EnSI %dx%dx%d, %d obs, max_points=%d" % (ny, nx, E, S, mp), "ms": t * 1e3,
                      "kernel_ms": gridpp.ensi_last_kernel_ms(), "Mcells/s": ny * nx / t / 1e6,
                      "GB/s_algorithmic": ny * nx * (8 * E + 16) / t / 1e9}), flush=True)


def host_case():
    """The same calls from numpy buffers (GPP_MEM_HOST): what a user of the reference's API sees, PCIe both ways included."""
    ny = nx = 4000
    lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(ny, nx, 10000, 1002, 0, ny)
    gridpp.Grid(lats[:300, :300], lons[:300, :300])          # first use of the device pays the runtime's start-up
    lats64, lons64 = lats.astype(np.float64), lons.astype(np.float64)
    t_grid = timeit(lambda: gridpp.Grid(lats64, lons64), reps=2, warm=0)
    grid = gridpp.Grid(lats, lons)
    lats32, lons32 = lats.astype(np.float32), lons.astype(np.float32)
    t_grid32 = timeit(lambda: gridpp.Grid(lats32, lons32), reps=2, warm=0)
    points = gridpp.Points(plat, plon)
    st = gridpp.BarnesStructure(10000)
    t = timeit(lambda: gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, 30))
    print(json.dumps({"case": "C3 OI 4000x4000 from numpy float32 buffers (64 MB in, 64 MB out over PCIe)", "ms": t * 1e3, "Mcells/s": ny * nx / t / 1e6,
                      "Grid_create_ms_float64_inputs": t_grid * 1e3, "Grid_create_ms_float32_inputs": t_grid32 * 1e3}), flush=True)
    E = 100
    cube = np.random.default_rng(3).random((1000, 4000, E), dtype=np.float32)      # a quarter of C4 (1.6 GB) to bound the host time
    t = timeit(lambda: gridpp.neighbourhood(cube, 15, gridpp.Mean), reps=2)
    print(json.dumps({"case": "C4/4 neighbourhood Mean 1000x4000x100 from a numpy float32 cube (1.6 GB over PCIe)", "ms": t * 1e3,
                      "GB/s_host_to_result": cube.nbytes / t / 1e9}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["oi", "nb", "ensi"]
    if "oi" in which:
        oi_case("C1 OI 200x200, 10 obs, mp=10", 200, 200, 10, 10, 1000)
        oi_case("C2 OI 1000x1000, 1k obs, mp=20", 1000, 1000, 1000, 20, 1001)
        oi_case("C3 OI 4000x4000, 10k obs, mp=30", 4000, 4000, 10000, 30, 1002)
        oi_case("C3 OI 4000x4000, 10k obs, mp=30, smooth terrain elev+laf (v=200,w=0.5)", 4000, 4000, 10000, 30, 1002, elev=True)
        oi_case("C3 OI 4000x4000, 10k obs, mp=30, white-noise elev+laf (v=200,w=0.5)", 4000, 4000, 10000, 30, 1002, elev="noise")
    if "nb" in which:
        nb_case(4000, 4000, 100, 15)
    if "host" in which:
        host_case()
    if "ensi" in which:
        ensi_case(500, 500, 50, 5000, 30)
        ensi_case(2500, 2500, 50, 5000, 30)
