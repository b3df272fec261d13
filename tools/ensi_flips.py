"""How many float32 outputs of optimal_interpolation_ensi differ from the oracle's, per seed of tools/ensi_hostile_soak.py (cached answers):
    python tools/ensi_flips.py SEED [SEED ...] [tile]
A value that differs by one float32 ulp of its operands shows up here long before it leaves the 1e-5 measure (A/B of accuracy changes)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
seeds = [int(a) for a in sys.argv[1:] if a.isdigit()]
sys.argv = [sys.argv[0], "0", "0", "0"] + [a for a in sys.argv[1:] if not a.isdigit()]
import tools.ensi_hostile_soak as S      # noqa: E402  (runs no pass: LO = HI = REPEATS = 0)
tot = dif = 0
for seed in seeds:
    c, ref = S.reference(seed)
    for conv in (False, True):
        S.gridpp.ensi_set_convergence(conv)
        out = S.call(c)
        S.gridpp.ensi_set_convergence(False)
        m = ~np.isnan(ref)
        e = np.abs(out[m].astype(np.float64) - ref[m]) / np.maximum(np.abs(ref[m]), 1e-2)
        nd = int((out[m] != ref[m]).sum())
        if not conv:
            tot += int(m.sum()); dif += nd
        print("seed %3d %-22s mp=%2d E=%2d %s: %6d values, %5d differ from the oracle's float, worst plain deviation %.3g" % (seed, c["kind"], c["mp"], c["E"], "converged" if conv else "default  ", int(m.sum()), nd, e.max() if e.size else 0), flush=True)
print("default mode: %d of %d values differ" % (dif, tot))
