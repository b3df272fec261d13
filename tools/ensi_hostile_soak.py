"""Hostile soak of optimal_interpolation_ensi against the CPU oracle (round-4 verdict, item 1).

    python tools/ensi_hostile_soak.py LO HI REPEATS [poison] [tile] [dump=DIR]

The kernels of this path keep HBM state between calls (csrc/ensi.hip: the park of k_ensi_pair, two Gram matrices per tile, parked
selections + metadata + signatures, packed observations, lists and counters), and k_ensi_pair warm-starts a cell from the PREVIOUS
cell's eigenvectors -- the history-dependent class of bug that DESIGN 9.1 describes for OI.  What is hostile here:
  * the oracle's answers are cached per seed, so REPEATS passes cost GPU time only;
  * every pass visits the seeds in another (seeded) order, with an unrelated larger call of another shape now and then in between,
    so that a workspace is reused at the same address with a different geometry (tile count, selection length, member count);
  * `poison` (tools/hostile/build.sh): before every call all LDS, 500 registers per lane and EVERY byte of the call-to-call
    workspaces are 0xFF (tools/hostile/harness.py);
  * every call is made twice and the results must agree bit for bit;
  * default mode (early-stopped sweeps + float32 perturbation series) and converged mode alternate from call to call, both held to the
    PLAIN measure |out - ref| / max(|ref|, 1e-2) < 1e-5;
  * the inputs are not the benign ones of tests/test_gpu_ensi_parity.py::case only: every third seed scales the observation sigmas
    (x 0.1 / 0.01), the member spread (x 0.01 / 100), moves the observations 20 spreads away or duplicates a member.
`tile` restricts max_points to 1..32 (the k_ensi_pair / k_ensi_members path only).
Prints one line per pass and `FAILURES n` (exit code 1 if n > 0)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1:]
lo, hi, repeats = int(args[0]), int(args[1]), int(args[2])
flags = args[3:]
from tools.hostile.harness import Hostile, same_bits, plain_mismatch, cached, ROOT      # noqa: E402
H = Hostile("poison" in flags)
gridpp = H.gridpp
from oracle import oracle as O                                                    # noqa: E402
from tests.test_gpu_ensi_parity import case                                       # noqa: E402
dump = next((f.split("=", 1)[1] for f in flags if f.startswith("dump=")), os.path.join(ROOT, "gpurun_out", "hostile_ensi"))
tile_only = "tile" in flags


def config(seed):
    rng = np.random.default_rng(770000 + seed)
    E = int(rng.choice([2, 5, 10, 17, 30, 50, 64, 70]))
    S = int(rng.choice([20, 60, 150, 300]))
    mp = int(rng.choice([1, 3, 10, 20, 30, 32] if tile_only else [0, 1, 3, 10, 30, 32, 40, 60]))
    h = float(rng.choice([15000.0, 30000.0, 60000.0]))
    Y, X = int(rng.integers(5, 20)), int(rng.integers(5, 20))
    lats, lons, bg, plat, plon, pbg, obs, sig = case(880000 + seed, Y, X, E, S, nan_member=(1 if seed % 5 == 0 and E > 2 else None), nan_obs=(seed % 7 == 0))
    kind = "benign"
    if seed % 3 == 1:
        k = int(rng.integers(0, 6))
        if k == 0:
            sig = (sig * 0.1).astype(np.float32); kind = "sigma x 0.1"
        elif k == 1:
            sig = (sig * 0.01).astype(np.float32); kind = "sigma x 0.01"
        elif k == 2:
            m = bg.mean(axis=2, keepdims=True); bg = (m + 0.01 * (bg - m)).astype(np.float32)
            pm = pbg.mean(axis=1, keepdims=True); pbg = (pm + 0.01 * (pbg - pm)).astype(np.float32); kind = "spread x 0.01"
        elif k == 3:
            m = bg.mean(axis=2, keepdims=True); bg = (m + 100 * (bg - m)).astype(np.float32)
            pm = pbg.mean(axis=1, keepdims=True); pbg = (pm + 100 * (pbg - pm)).astype(np.float32); kind = "spread x 100"
        elif k == 4:
            obs = (obs + 20 * np.where(rng.random(S) < 0.5, -1, 1)).astype(np.float32); kind = "obs +-20 spreads"
        elif E > 2:
            bg[:, :, E - 1] = bg[:, :, 0] * np.float32(1 + 1e-6); pbg[:, E - 1] = pbg[:, 0] * np.float32(1 + 1e-6); kind = "near-duplicate members"
    elev = seed % 4 == 0
    return dict(lats=lats, lons=lons, bg=bg, plat=plat, plon=plon, pbg=pbg, obs=obs, sig=sig, h=h, mp=mp, allow=bool(seed % 2), v=(200.0 if elev else 0.0), elev=elev,
                Y=Y, X=X, E=E, S=S, kind=kind, ge=np.random.default_rng(99).uniform(0, 500, (Y, X)) if elev else (), pe=np.random.default_rng(98).uniform(0, 500, S) if elev else ())


refs = {}


def reference(seed):
    if seed not in refs:
        c = config(seed)

        def compute():
            og = O.Pts(c["lats"].ravel(), c["lons"].ravel(), c["ge"].ravel() if c["elev"] else None)
            op = O.Pts(c["plat"], c["plon"], c["pe"] if c["elev"] else None)
            return dict(ref=O.oi_ensi(og, c["bg"].reshape(-1, c["E"]), op, c["obs"], c["sig"], c["pbg"], O.Barnes(c["h"], c["v"], 0), c["mp"], c["allow"]).reshape(c["Y"], c["X"], c["E"]))
        refs[seed] = (c, cached("ensi_tile" if tile_only else "ensi", seed, c, compute)["ref"])
    return refs[seed]


failures = []
worst = {"default": 0.0, "converged": 0.0}
by_kind = {}


def record(seed, what, detail, c, arrays):
    failures.append((seed, what, detail))
    print("FAIL seed %d [%s]: %s; Y=%d X=%d E=%d S=%d h=%g mp=%d allow=%d %s" % (seed, what, detail, c["Y"], c["X"], c["E"], c["S"], c["h"], c["mp"], c["allow"], c["kind"]), flush=True)
    os.makedirs(dump, exist_ok=True)
    np.savez(os.path.join(dump, "fail_%d_%s_%d.npz" % (seed, what.replace(" ", "_"), len(failures))), **arrays, **{k: v for k, v in c.items() if isinstance(v, np.ndarray)})


def call(c):
    grid = gridpp.Grid(c["lats"], c["lons"], c["ge"], ())
    points = gridpp.Points(c["plat"], c["plon"], c["pe"], ())
    H.before_call()
    return np.asarray(gridpp.optimal_interpolation_ensi(grid, c["bg"], points, c["obs"], c["sig"], c["pbg"], gridpp.BarnesStructure(c["h"], c["v"], 0), c["mp"], c["allow"]))


def one(seed, converged):
    c, ref = reference(seed)
    mode = "converged" if converged else "default"
    gridpp.ensi_set_convergence(converged)
    try:
        out = call(c)
        again = call(c)
    finally:
        gridpp.ensi_set_convergence(False)
    d = plain_mismatch(out, ref, 1e-2)
    if d:
        record(seed, mode, d, c, dict(out=out, ref=ref))
    else:
        m = ~np.isnan(ref)
        if m.any():
            e = float((np.abs(out[m].astype(np.float64) - ref[m]) / np.maximum(np.abs(ref[m]), 1e-2)).max())
            worst[mode] = max(worst[mode], e)
            by_kind[(c["kind"], mode)] = max(by_kind.get((c["kind"], mode), 0.0), e)
    if not same_bits(out, again):
        record(seed, mode + " repeat differs", "%d values" % int((out.view(np.uint32) != again.view(np.uint32)).sum()), c, dict(out=out, out_again=again, ref=ref))


def unrelated(rng):
    """a larger call of another shape in between: the workspaces grow or are reused with another tile count / member count"""
    Y, X, E, S = int(rng.integers(30, 90)), int(rng.integers(30, 90)), int(rng.choice([8, 24, 50])), int(rng.integers(200, 900))
    lats, lons, bg, plat, plon, pbg, obs, sig = case(int(rng.integers(1 << 30)), Y, X, E, S)
    H.before_call()
    gridpp.optimal_interpolation_ensi(gridpp.Grid(lats, lons), bg, gridpp.Points(plat, plon), obs, sig, pbg, gridpp.BarnesStructure(float(rng.choice([8000.0, 20000.0]))), int(rng.choice([5, 30, 45, 0])), True)


def main():
    t0 = time.time()
    for seed in range(lo, hi):
        reference(seed)
    print("oracle answers for seeds %d..%d in %.0f s%s" % (lo, hi, time.time() - t0, " (tile path only)" if tile_only else ""), flush=True)
    for rep in range(repeats):
        rng = np.random.default_rng(4242 + rep)
        order = np.arange(lo, hi) if rep == 0 else rng.permutation(np.arange(lo, hi))
        t1, nf = time.time(), len(failures)
        for k, seed in enumerate(order):
            if rep > 0 and k % 29 == 3:
                unrelated(rng)
            one(int(seed), converged=bool((k + rep) & 1))
        print("pass %d (%s order%s): %d failures in %.0f s; worst plain deviation so far: default %.3g, converged %.3g" % (
            rep, "sequential" if rep == 0 else "shuffled", ", poisoned" if H.poison else "", len(failures) - nf, time.time() - t1, worst["default"], worst["converged"]), flush=True)
    print("worst plain deviation by input kind: " + "; ".join("%s / %s %.3g" % (k[0], k[1], v) for k, v in sorted(by_kind.items())))
    print("library calls: %d" % H.calls)
    print("FAILURES %d" % len(failures))
    for f in failures[:20]:
        print(f)
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
