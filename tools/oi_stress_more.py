"""More seeds of tests/test_gpu_oi_union_stress.py::test_random_configurations than the test suite runs (a one-off soak)."""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_oi_union_stress import _random_configuration
mps = [33, 40, 50, 62] if len(sys.argv) > 3 and sys.argv[3] == "62" else [1, 2, 7, 20, 30, 32]
one = lambda seed: _random_configuration(seed, mps)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); bad = []
for seed in range(lo, hi):
    try:
        one(seed)
    except AssertionError as e:
        bad.append((seed, str(e)[:200], traceback.format_exc().splitlines()[-3].strip()[:150]))
print("seeds %d..%d: %d failures in %.0f s" % (lo, hi, len(bad), time.time() - t0))
for b in bad[:10]:
    print(b)
