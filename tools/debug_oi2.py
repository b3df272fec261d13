import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gridpp_amd as gridpp
from oracle import oracle as O
from tests.test_gpu_oi_parity import make_case
c = make_case(1001, 64, 64, 200)
og = O.Pts(c["lats"].ravel(), c["lons"].ravel()); op = O.Pts(c["plat"], c["plon"])
st = gridpp.BarnesStructure(10000); ost = O.Barnes(10000)
cell = 36*64+60
nd = 0
for s in range(200):
    p1 = gridpp.Point(0,0,np.nan,np.nan,gridpp.Geodetic, og.x[cell], og.y[cell], og.z[cell])
    p2 = gridpp.Point(0,0,np.nan,np.nan,gridpp.Geodetic, op.x[s], op.y[s], op.z[s])
    a = st.corr(p1, p2)
    b = ost.corr((og.x[cell], og.y[cell], og.z[cell], np.nan, np.nan), (op.x[s], op.y[s], op.z[s], np.nan, np.nan))
    if np.float32(a) != np.float32(b):
        nd += 1
        print("rho differs", s, repr(a), repr(b))
print("rho mismatches:", nd)
sel, tie = O.oi_selection(og, cell, op, c["obs"], c["pbg"], ost, 20)
print("oracle sel", sel, tie)
