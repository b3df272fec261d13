"""Smooth-terrain OI case (elevation + laf dependent rho on a synthetic topography) at a chosen grid size: timing / rocprofv3 runs."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_cases import oi_case
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
S = int(10000 * (n / 4000.0) ** 2)
print(json.dumps(oi_case("OI %dx%d, %d obs, mp=30, smooth terrain elev+laf" % (n, n, S), n, n, S, 30, 1002, elev=True, reps=3)))
