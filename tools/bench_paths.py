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


from tools.bench_cases import make_workload, timeit, oi_case as _oi_case, nb_cases, ensi_case as _ensi_case  # noqa: E402


def oi_case(*a, **k):
    print(json.dumps(_oi_case(*a, **k)), flush=True)


def nb_case(ny, nx, E, hw):
    for r in nb_cases(ny, nx, E, hw, qs=(0.5, 0.9)):
        print(json.dumps(r), flush=True)


def ensi_case(*a, **k):
    print(json.dumps(_ensi_case(*a, **k)), flush=True)


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
