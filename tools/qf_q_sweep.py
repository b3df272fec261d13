"""quantile_fast on config 4 (4000 x 4000 x 100, uniform [0, 10) members, thresholds 0..10, halfwidth 15) against the quantile: 0.5 and 0.9 sit ON a
threshold's expected rank (the window means of that plane scatter around the quantile itself), 0.55 / 0.97 between two.   python tools/qf_q_sweep.py [q ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import gridpp_amd as gridpp
from bench_cases import c4_cube, timeit
cube = c4_cube(4000, 4000, 100)
thr = torch.linspace(0, 10, 11, device="cuda")
for q in [float(a) for a in sys.argv[1:]] or [0.5, 0.55, 0.9, 0.97, 0.05, 0.3]:
    t = timeit(lambda: gridpp.neighbourhood_quantile_fast(cube, q, 15, thr), reps=5)
    print("q = %-5g %.3f ms" % (q, t * 1e3), flush=True)
