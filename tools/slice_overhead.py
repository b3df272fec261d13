"""Per-step cost of one rank's share of the headline workload on an N-GPU run (row tile of 4000/N rows), measured on one
GPU: wall time per step vs kernel time -> the fixed per-call overhead that limits strong scaling."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gridpp_amd as gridpp
from bench import make_workload
base = base_a = None
for N in (1, 2, 4, 8):
    rows = 4000 // N
    lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(4000, 4000, 10000, 1002, 0, rows)
    grid, points, st = gridpp.Grid(lats, lons), gridpp.Points(plat, plon), gridpp.BarnesStructure(10000)
    d = [torch.from_numpy(a).cuda() for a in (bg, obs, ratios, pbg)]
    for _ in range(3): gridpp.optimal_interpolation(grid, d[0], points, d[1], d[2], d[3], st, 30)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 30
    for _ in range(K): gridpp.optimal_interpolation(grid, d[0], points, d[1], d[2], d[3], st, 30)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K * 1e3
    s = gridpp.oi_last_stats()
    base = base or dt
    # the same steps with one analysis ahead (gpp_optimal_interpolation_full + GPP_ASYNC, gpp_wait: what bench.py --gpus N does)
    from gridpp_amd.dist import AnalysisPipeline
    pipe = AnalysisPipeline(1)
    for _ in range(3): pipe.push(gridpp.optimal_interpolation_async(grid, d[0], points, d[1], d[2], d[3], st, 30))
    pipe.drain()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): pipe.push(gridpp.optimal_interpolation_async(grid, d[0], points, d[1], d[2], d[3], st, 30))
    pipe.drain()
    torch.cuda.synchronize(); dta = (time.perf_counter() - t0) / K * 1e3
    base_a = base_a or dta
    print("N=%d rows=%d: %.3f ms/step, kernels %.3f ms (first pass %.3f), overhead %.3f ms, speed-up bound %.2f, declined tiles %d | one analysis ahead: %.3f ms/step, speed-up bound %.2f" % (
        N, rows, dt, s["kernel_ms"], s["union_kernel_ms"], dt - s["kernel_ms"], base / dt, s["fallback_tiles"], dta, base_a / dta))
