"""Large-n EnSI (k_ensi_big_ns: Newton-Schulz) on ill-conditioned Pinv: observation sigmas scaled down by 1e-1 .. 1e-6 (condition numbers up to
~1e13) against the oracle; cells the iteration gives up on go to k_ensi_huge (Jacobi).  Prints the worst deviation per scale."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_ensi_parity import case, run
from tests.ensi_golden import rel_err
for E in (20, 50):
    for scale in (1e-1, 1e-2, 1e-3, 1e-4, 1e-6):
        c = list(case(4242 + E, 6, 7, E, 60))
        c[7] = (c[7] * scale).astype(np.float32)
        out, ref = run(tuple(c), 200000, 0)
        err = rel_err(out, ref, c[2])
        print("E=%d sigma x %g: worst deviation %.2e (nan pattern equal: %s)" % (E, scale, np.nanmax(err), (np.isnan(out) == np.isnan(ref)).all()), flush=True)
