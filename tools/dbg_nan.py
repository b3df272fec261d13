import sys; sys.path.insert(0,'/root/repo')
import numpy as np, collections
import gridpp_amd as gridpp
from bench import make_workload
lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(4000,4000,10000,1002,0,4000)
grid = gridpp.Grid(lats, lons); points = gridpp.Points(plat, plon)
out = gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, gridpp.BarnesStructure(10000), 30)
bad = ~np.isfinite(out)
print("bad", bad.sum(), "nan", np.isnan(out).sum(), "inf", np.isinf(out).sum())
t = bad.reshape(500,8,500,8).transpose(0,2,1,3).reshape(500,500,64)
nb = t.sum(axis=2)
print("tiles with bad", (nb>0).sum(), "hist of bad per tile", collections.Counter(nb[nb>0].tolist()).most_common(12))
lanes = t.sum(axis=(0,1)); print("per lane", lanes.tolist())
ty, tx = np.nonzero((nb>0)&(nb<64))
for k in range(5):
    print((ty[k],tx[k]), np.nonzero(t[ty[k],tx[k]])[0].tolist())
    print(out[ty[k]*8:ty[k]*8+8, tx[k]*8:tx[k]*8+8])
