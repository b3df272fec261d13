import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gridpp_amd as gridpp
ny = nx = 4000
lats, lons = np.meshgrid(np.linspace(0, 1, ny), np.linspace(0, 1, nx), indexing='ij')
for i in range(3):
    t0 = time.perf_counter(); g = gridpp.Grid(lats, lons); t1 = time.perf_counter()
    print("Grid(4000x4000) create: %.1f ms" % ((t1 - t0) * 1e3))
bg = np.zeros((ny, nx), np.float32)
p = gridpp.Points(np.random.rand(10000), np.random.rand(10000))
for i in range(2):
    t0 = time.perf_counter(); v = gridpp.nearest(g, p, bg); t1 = time.perf_counter(); print("nearest grid->10k points: %.1f ms" % ((t1 - t0) * 1e3))
