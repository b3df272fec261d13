"""More seeds of tests/test_gpu_oi_union_stress.py::test_random_configurations_rough_terrain (white-noise elevations / land fractions,
elevation and laf dependent rho, against the oracle)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_oi_union_stress import test_random_configurations_rough_terrain as one
lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); bad = []
for seed in range(lo, hi):
    try:
        one(seed)
    except AssertionError as e:
        bad.append((seed, str(e)[:200]))
print("seeds %d..%d: %d failures in %.0f s" % (lo, hi, len(bad), time.time() - t0))
for b in bad[:10]:
    print(b)
