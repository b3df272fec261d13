"""White-noise terrain case (every cell its own observation set) at a chosen grid size: timing of the per-selection path."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_cases import oi_case
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
S = int(10000 * (n / 4000.0) ** 2)
print(json.dumps(oi_case("OI %dx%d, %d obs, mp=30, white-noise elev+laf" % (n, n, S), n, n, S, 30, 1002, elev="noise", reps=2)))
