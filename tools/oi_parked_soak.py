"""More seeds of tests/test_gpu_oi_union_stress.py::test_parked_selections_rough_terrain (k_oi parks the single-member groups, k_oi_pairs
solves them: analysis and variance against the oracle, and bit-identical to k_oi solving them itself)."""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_oi_union_stress import test_parked_selections_rough_terrain as one


import gridpp_amd


class Env:
    """stands in for pytest's monkeypatch OUTSIDE pytest: the library reads no environment variable (round 4) and no conftest forwards the
    process environment here, so the switch goes to the library's own hook (writing os.environ after the library was loaded
    never reached it, and the NO_PAIRS leg of the test ran the same path as the first call)"""
    def setenv(self, k, v):
        os.environ[k] = v
        gridpp_amd.set_path_override(k, v)


lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); bad = []; skipped = 0
for seed in range(lo, hi):
    try:
        one(seed, Env())
    except BaseException as e:
        if type(e).__name__ == "Skipped":
            skipped += 1
            continue
        if not isinstance(e, AssertionError):
            raise
        bad.append((seed, str(e)[:200], traceback.format_exc().splitlines()[-3].strip()[:160]))
    finally:
        os.environ.pop("GPP_OI_NO_PAIRS", None)
        gridpp_amd.set_path_override("GPP_OI_NO_PAIRS", None)
print("seeds %d..%d: %d failures, %d skipped (geometry keeps the first pass) in %.0f s" % (lo, hi, len(bad), skipped, time.time() - t0))
for b in bad[:10]:
    print(b)
