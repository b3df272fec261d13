"""Hostile soak of optimal_interpolation_ensi_multi_{ebe, ebesc, utem} against the CPU oracle (round-4 verdict, item 1).

    python tools/ensi_multi_hostile_soak.py LO HI REPEATS [poison] [dump=DIR]

The three filters share the workspace of optimal_interpolation_ensi (csrc/ensi.hip: packed observations, member flags, the Y
matrices, candidate keys, lists, counters, the HBM scratch of k_ensi_multi_huge).  Cached oracle answers, shuffled order, an EnSI call
of another shape now and then in between (the SAME buffers with another meaning), every call twice with bit-identical results, and
with `poison` LDS, registers and every workspace byte 0xFF before each call (tools/hostile/harness.py).  Sizes on both sides of the
LDS areas of k_ensi_multi (64 selected observations / 64 valid members), an invalid last member in every fourth configuration.
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
dump = next((f.split("=", 1)[1] for f in flags if f.startswith("dump=")), os.path.join(ROOT, "gpurun_out", "hostile_ensi_multi"))


def config(seed):
    rng = np.random.default_rng(550000 + seed)
    variant = ["ebe", "ebesc", "utem"][seed % 3]
    n = int(rng.integers(20, 260))
    E = int(rng.choice([3, 8, 12, 30, 64, 70] if variant == "utem" else [3, 8, 12, 30]))
    S = int(rng.choice([15, 50, 120]))
    mp = int(rng.choice([0, 5, 20, 64, 90]))
    h = float(rng.choice([15000.0, 30000.0, 60000.0]))
    f32 = np.float32
    blat, blon = rng.random(n).astype(f32), rng.random(n).astype(f32)
    plat, plon = rng.random(S).astype(f32), rng.random(S).astype(f32)
    bg, bgc = rng.normal(0, 1, (n, E)).astype(f32), rng.normal(0, 1, (n, E)).astype(f32)
    pbg, pbgc = rng.normal(0, 1, (S, E)).astype(f32), rng.normal(0, 1, (S, E)).astype(f32)
    if seed % 4 == 3 and E > 3:
        bg[:, E - 1] = np.nan                      # an invalid LAST member stays untouched (one in front of a valid one raises: DESIGN 2)
    pobs = rng.normal(0, 1, S).astype(f32) if variant == "utem" else rng.normal(0, 1, (S, E)).astype(f32)
    if seed % 5 == 2:
        (pobs if variant == "utem" else pobs[:, 0])[::6] = np.nan
    pr, br = rng.uniform(0.1, 1.5, S).astype(f32), rng.uniform(0.5, 1.5, n).astype(f32)
    return dict(variant=variant, n=n, E=E, S=S, mp=mp, h=h, allow=bool(seed % 2), blat=blat, blon=blon, plat=plat, plon=plon, bg=bg, bgc=bgc, pbg=pbg, pbgc=pbgc,
                pobs=pobs, pr=pr, br=br)


refs = {}


def reference(seed):
    if seed not in refs:
        c = config(seed)
        ref = cached("ensi_multi", seed, c, lambda: dict(ref=O.oi_ensi_multi(c["variant"], O.Pts(c["blat"], c["blon"]), c["br"], c["bg"], c["bgc"], O.Pts(c["plat"], c["plon"]),
                                                                             c["pobs"], c["pr"], c["pbg"], c["pbgc"], O.Barnes(c["h"]), c["mp"], c["allow"])))["ref"]
        refs[seed] = (c, ref)
    return refs[seed]


failures = []
worst = 0.0


def record(seed, what, detail, c, arrays):
    failures.append((seed, what, detail))
    print("FAIL seed %d [%s %s]: %s; n=%d E=%d S=%d h=%g mp=%d allow=%d" % (seed, c["variant"], what, detail, c["n"], c["E"], c["S"], c["h"], c["mp"], c["allow"]), flush=True)
    os.makedirs(dump, exist_ok=True)
    np.savez(os.path.join(dump, "fail_%d_%s_%d.npz" % (seed, what.replace(" ", "_"), len(failures))), **arrays, **{k: v for k, v in c.items() if isinstance(v, np.ndarray)})


def call(c):
    b, p, st = gridpp.Points(c["blat"], c["blon"]), gridpp.Points(c["plat"], c["plon"]), gridpp.BarnesStructure(c["h"])
    H.before_call()
    if c["variant"] == "ebe":
        out = gridpp.optimal_interpolation_ensi_multi_ebe(b, c["br"], c["bg"], c["bgc"], p, c["pobs"], c["pr"], c["pbg"], c["pbgc"], st, c["mp"], c["allow"])
    elif c["variant"] == "ebesc":
        out = gridpp.optimal_interpolation_ensi_multi_ebesc(b, c["br"], c["bg"], p, c["pobs"], c["pr"], c["pbg"], st, c["mp"], c["allow"])
    else:
        out = gridpp.optimal_interpolation_ensi_multi_utem(b, c["br"], c["bg"], c["bgc"], p, c["pobs"], c["pr"], c["pbg"], c["pbgc"], st, c["mp"], c["allow"])
    return np.asarray(out)


def one(seed):
    global worst
    c, ref = reference(seed)
    out = call(c)
    again = call(c)
    d = plain_mismatch(out, ref, 1e-2)
    if d:
        record(seed, "analysis", d, c, dict(out=out, ref=ref))
    else:
        m = ~np.isnan(ref)
        if m.any():
            worst = max(worst, float((np.abs(out[m].astype(np.float64) - ref[m]) / np.maximum(np.abs(ref[m]), 1e-2)).max()))
    if not same_bits(out, again):
        record(seed, "repeat differs", "%d values" % int((out.view(np.uint32) != again.view(np.uint32)).sum()), c, dict(out=out, out_again=again, ref=ref))


def unrelated(rng):
    Y, X, E, S = int(rng.integers(20, 60)), int(rng.integers(20, 60)), int(rng.choice([8, 24, 50])), int(rng.integers(100, 600))
    lats, lons, bg, plat, plon, pbg, obs, sig = case(int(rng.integers(1 << 30)), Y, X, E, S)
    H.before_call()
    gridpp.optimal_interpolation_ensi(gridpp.Grid(lats, lons), bg, gridpp.Points(plat, plon), obs, sig, pbg, gridpp.BarnesStructure(float(rng.choice([8000.0, 20000.0]))), int(rng.choice([5, 30, 45, 0])), True)


t0 = time.time()
for seed in range(lo, hi):
    reference(seed)
print("oracle answers for seeds %d..%d in %.0f s" % (lo, hi, time.time() - t0), flush=True)
for rep in range(repeats):
    rng = np.random.default_rng(999 + rep)
    order = np.arange(lo, hi) if rep == 0 else rng.permutation(np.arange(lo, hi))
    t1, nf = time.time(), len(failures)
    for k, seed in enumerate(order):
        if rep > 0 and k % 23 == 4:
            unrelated(rng)
        one(int(seed))
    print("pass %d (%s order%s): %d failures in %.0f s; worst plain deviation so far %.3g" % (rep, "sequential" if rep == 0 else "shuffled", ", poisoned" if H.poison else "", len(failures) - nf, time.time() - t1, worst), flush=True)
print("library calls: %d" % H.calls)
print("FAILURES %d" % len(failures))
for f in failures[:20]:
    print(f)
sys.exit(1 if failures else 0)
