import sys, json, os
sys.path.insert(0, '/root/repo')
import numpy as np
import gridpp_amd as g
G = json.load(open('/root/repo/tests/golden/reference_known_answers.json')) if os.path.exists('/root/repo/tests/golden/reference_known_answers.json') else None
if G is None:
    import glob; print(glob.glob('/root/repo/tests/golden/*')[:20]); sys.exit(0)
e = G["oi_extrapolation"]
a, b, n = e["grid_y_linspace"]
y = np.linspace(a, b, n); x = np.zeros(n)
grid = g.Points(y, x, x, x, g.Cartesian)
py = e["points_y"]; z4 = [0] * len(py)
points = g.Points(py, z4, z4, z4, g.Cartesian)
pr = e["pratio"] * np.ones(len(py))
st = g.BarnesStructure(e["h"])
o0 = np.asarray(g.optimal_interpolation(grid, np.zeros(n), points, e["pobs"], pr, np.zeros(len(py)), st, e["max_points"], False))
o1 = np.asarray(g.optimal_interpolation(grid, np.zeros(n), points, e["pobs"], pr, np.zeros(len(py)), st, e["max_points"], True))
print("n", n, "a b", a, b, "py", py, "h", e["h"], "mp", e["max_points"])
print("max o0", o0.max(), "max o1", o1.max())
np.save(os.environ.get("OUTF", "/tmp/o.npy"), np.stack([o0, o1]))
print(g.oi_last_stats())
