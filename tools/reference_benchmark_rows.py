"""The rows of the reference's own benchmark script (tests/benchmark.py) that lie on this path, with the same shapes and
the same kind of timing: wall time around the Python call with numpy inputs, i.e. INCLUDING host<->device copies (the
reference's figures include its numpy<->std::vector copies).  "expected" = the reference's hard-coded single-core figure
(Intel i7 3.40 GHz, tests/benchmark.py:52-66,90)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gridpp_amd as gridpp

np.random.seed(1000)


def grid(n):
    y, x = np.meshgrid(np.linspace(0, 1, n), np.linspace(0, 1, n))
    return gridpp.Grid(y, x, 0 * x, 0 * x)


def points(n):
    return gridpp.Points(np.linspace(0, 1, n), np.linspace(0, 1, n), np.zeros(n), np.zeros(n))


def timeit(f, reps=3):
    f()                       # the first call of a kind pays one-off costs (module load, index builds of new objects stay in)
    ts = []
    for _ in range(reps):
        t0 = time.time(); f(); ts.append(time.time() - t0)
    return min(ts)


I1000, I2000 = np.zeros((1000, 1000)), np.zeros((2000, 2000))
structure = gridpp.BarnesStructure(10000)
rows = []
y1000, x1000 = np.meshgrid(np.linspace(0, 1, 1000), np.linspace(0, 1, 1000))
rows.append(("Grid 1000^2", 0.74, lambda: gridpp.Grid(y1000, x1000, 0 * x1000, 0 * x1000)))
Z10000 = np.zeros((10000, 10000))
rows.append(("neighbourhood 10000^2 hw=7 Mean", 2.05, lambda: gridpp.neighbourhood(Z10000, 7, gridpp.Mean)))
rows.append(("neighbourhood 2000^2 hw=7 Max", 0.99, lambda: gridpp.neighbourhood(I2000, 7, gridpp.Max)))
rows.append(("neighbourhood_quantile_fast 2000^2 hw=7, 11 thresholds", 1.23, lambda: gridpp.neighbourhood_quantile_fast(I2000, 0.5, 7, np.linspace(0, 1, 11))))
I500 = np.zeros((500, 500))
rows.append(("neighbourhood_quantile 500^2 hw=7", 1.70, lambda: gridpp.neighbourhood_quantile(I500, 0.5, 7)))
G1000 = grid(1000)
rows.append(("nearest 1000^2 grid -> grid", 1.52, lambda: gridpp.nearest(G1000, G1000, I1000)))
rows.append(("bilinear 1000^2 grid -> grid", 1.68, lambda: gridpp.bilinear(G1000, G1000, I1000)))
I50 = np.zeros((50, 1000, 1000))
rows.append(("bilinear 1000^2 x 50 levels", 4.42, lambda: gridpp.bilinear(G1000, G1000, I50)))
G200, P100000 = grid(200), points(100000)
V100000 = np.zeros(100000)
rows.append(("gridding 200^2, 100000 points, radius 5000, Mean", 0.61, lambda: gridpp.gridding(G200, P100000, V100000, 5000, 1, gridpp.Mean)))
rows.append(("gridding_nearest 200^2, 100000 points, Mean", 0.11, lambda: gridpp.gridding_nearest(G200, P100000, V100000, 1, gridpp.Mean)))
rows.append(("nearest 1000^2 x 50 levels", 1.93, lambda: gridpp.nearest(G1000, G1000, I50)))
Z200 = np.zeros((200, 200))
rows.append(("fill 200^2, 100000 points, radius 5000", 1.96, lambda: gridpp.fill(G200, Z200, P100000, np.ones(100000) * 5000, 1, False)))
rows.append(("doping_square 200^2, 100000 points, halfwidth 5", 0.12, lambda: gridpp.doping_square(G200, Z200, P100000, np.ones(100000), np.ones(100000, "int") * 5, False)))
rows.append(("doping_circle 200^2, 100000 points, radius 5000", 2.00, lambda: gridpp.doping_circle(G200, Z200, P100000, np.ones(100000), np.ones(100000) * 5000, False)))
R2000a, R2000b = np.random.rand(2000, 2000) * 100, np.random.rand(2000, 2000)
rows.append(("calc_gradient 2000^2 LinearRegression hw=10", 0.45, lambda: gridpp.calc_gradient(R2000a, I2000, gridpp.LinearRegression, 10, 0, 100, 0)))
A2000 = np.random.rand(2000, 2000) < 0.5
rows.append(("neighbourhood_search 2000^2 7x7", 1.11, lambda: gridpp.neighbourhood_search(R2000b, R2000b, 3, 0.7, 1, 0.1, A2000)))
G100, P1000 = grid(100), points(1000)
rows.append(("optimal_interpolation 100^2, 1000 obs, max_points 20", 0.80,
             lambda: gridpp.optimal_interpolation(G100, np.zeros((100, 100)), P1000, np.zeros(1000), np.ones(1000), np.ones(1000), structure, 20)))
sgrid = gridpp.BarnesStructure(G100, np.full((100, 100), 10000.0), np.zeros((100, 100)), np.zeros((100, 100)))
rows.append(("optimal_interpolation, spatially varying length scale", 0.91,
             lambda: gridpp.optimal_interpolation(G100, np.zeros((100, 100)), P1000, np.zeros(1000), np.ones(1000), np.ones(1000), sgrid, 20)))
for name, expected, f in rows:
    t = timeit(f)
    print(json.dumps({"row": name, "reference_expected_s_1core_i7": expected, "this_s": round(t, 5), "ratio": round(expected / t, 1)}), flush=True)
