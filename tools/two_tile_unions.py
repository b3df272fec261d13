"""Round-4 verdict, item 4: "one factorisation per TWO tiles ... measure the declined-tile rate first".  On the CPU (numpy + scipy, no GPU):
for a sample of tiles of the headline workload (4000 x 4000 grid, 10 000 observations, BarnesStructure(10000), max_points 30; tiles of 8 x 8
cells as k_oi_union cuts them) the size of the union of the cells' selections, its core (selected by every cell) and the extras per cell --
for one tile and for the pair of it and its right-hand neighbour (8 x 16 cells) -- against the limits of the shared factorisation
(csrc/oi_union.h: union <= 40 rows, <= 12 extras in all, <= 6 per cell).  Selection = the max_points nearest observations inside the
localization radius (rho is monotone in the distance for this structure; src/api/oi.cpp:229-273)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.spatial import cKDTree
from tools.bench_cases import make_workload

ny = nx = 4000
S, mp, h = 10000, 30, 10000.0
R = h * np.sqrt(-2 * np.log(0.0013))            # the radius at which Barnes' rho falls below its cut (structure.cpp:26-34: hmax)
lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(ny, nx, S, 1002, 0, ny)
def xyz(la, lo):
    la, lo = np.radians(la), np.radians(lo)
    return 6.371e6 * np.stack([np.cos(la) * np.cos(lo), np.cos(la) * np.sin(lo), np.sin(la)], axis=-1)
tree = cKDTree(xyz(plat, plon))
rng = np.random.default_rng(5)
def stats(cells_lat, cells_lon):
    d, idx = tree.query(xyz(cells_lat.ravel(), cells_lon.ravel()), k=mp, distance_upper_bound=R)
    sels = [set(i[np.isfinite(dd)]) for dd, i in zip(d, idx)]
    union = set().union(*sels)
    core = set.intersection(*sels) if sels else set()
    extras = max(len(s - core) for s in sels)
    return len(union), len(core), len(union) - len(core), extras
fits = lambda u, c, e, m: u <= 40 and e <= 12 and m <= 6
one, two = [], []
for _ in range(400):
    ty, tx = int(rng.integers(0, ny // 8)), int(rng.integers(0, nx // 16)) * 2
    ys = slice(8 * ty, 8 * ty + 8)
    one.append(stats(lats[ys, 8 * tx:8 * tx + 8], lons[ys, 8 * tx:8 * tx + 8]))
    two.append(stats(lats[ys, 8 * tx:8 * tx + 16], lons[ys, 8 * tx:8 * tx + 16]))
for name, a in (("one tile (8 x 8 cells)", np.array(one)), ("two tiles (8 x 16 cells)", np.array(two))):
    ok = np.mean([fits(*r) for r in a])
    print("%-26s union %.1f (max %d), core %.1f, extras in all %.1f (max %d), extras of the worst cell %.1f (max %d): %.1f %% fit the shared factorisation" % (
        name, a[:, 0].mean(), a[:, 0].max(), a[:, 1].mean(), a[:, 2].mean(), a[:, 2].max(), a[:, 3].mean(), a[:, 3].max(), 100 * ok))
