"""Hostile soak of the randomised OI parity test (tests/test_gpu_oi_union_stress.py::_random_inputs) against the CPU oracle.

    python tools/oi_hostile_soak.py LO HI REPEATS [62] [poison] [reuse] [spatial] [pairs] [dump=DIR]

What is hostile about it (round-3 verdict, item 1: one unreproduced failure of the max_points 33..62 sequence):
  * the oracle's answers are computed once per seed and cached, so REPEATS passes over the sequence cost GPU time only;
  * every pass visits the seeds in a different (seeded) random order, with an unrelated large call now and then in between;
  * `poison` (needs tools/hostile/build.sh): before every call all 160 KB of LDS of every CU, 500 registers per lane of every SIMD and
    every byte of the library's call-to-call OI workspaces (lists, counters, parked selections, packed observations) are filled with
    0xFF -- a kernel that reads what this call never wrote meets NaNs, negative list entries and absurd counts;
  * every call is made twice and the two results must agree bit for bit (a race shows up as a difference even when both are within
    tolerance);
  * `reuse` (round 5): the Grid / Points handles of a seed are kept from pass to pass, so that the state a GEOMETRY carries between calls is
    exercised under the same hostility -- the memory of the tiles its first pass declined (list + flag bytes in HBM, csrc/oi.hip `overlap`), the
    list passes beside the first pass on the second stream, and the third call of a seed goes through GPP_ASYNC + gpp_wait (deferred on three
    streams when the geometry is in its steady state);
  * a failure records WHICH check failed, the statistics of the call, and dumps inputs + both outputs + the oracle's to DIR.
Prints one summary line per pass and `FAILURES n` at the end (exit code 1 if n > 0).
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
lo, hi, repeats = int(args[0]), int(args[1]), int(args[2])
flags = args[3:]
mps = [33, 40, 50, 62] if "62" in flags else [1, 2, 7, 20, 30, 32]
poison = "poison" in flags
reuse = "reuse" in flags
spatial = "spatial" in flags     # round 6: additionally a spatially varying Barnes structure on the same inputs (k_oi_union_sp + its list passes)
if "pairs" in flags:             # round 6: the first pass with one factorisation per pair of tiles (k_oi_union_pair, behind its switch)
    os.environ["GPP_OI_PAIR_TILES"] = "1"
dump = next((f.split("=", 1)[1] for f in flags if f.startswith("dump=")), os.path.join(ROOT, "gpurun_out", "hostile"))
from tools.hostile.harness import Hostile                      # noqa: E402  (sets GPP_LIB for the poisoned build before gridpp_amd loads)
H = Hostile(poison)
gridpp = H.gridpp
from oracle import oracle as O                                  # noqa: E402
from tests.test_gpu_oi_union_stress import _random_inputs       # noqa: E402

RTOL = 1e-5


def hostile():
    H.before_call()


def compare(out, ref):
    """None when `out` matches the oracle's `ref`, otherwise what differs"""
    if out.shape != ref.shape:
        return "shape %s vs %s" % (out.shape, ref.shape)
    dn = np.isnan(out) != np.isnan(ref)
    if dn.any():
        return "NaN pattern differs in %d cells, first at %s" % (int(dn.sum()), np.argwhere(dn)[0].tolist())
    m = ~np.isnan(ref)
    if m.any():
        err = np.abs(out[m].astype(np.float64) - ref[m]) / np.maximum(np.abs(ref[m]), 1e-3)
        if not err.max() < RTOL:
            return "max rel err %.3g (%d cells above %.0e)" % (err.max(), int((err >= RTOL).sum()), RTOL)
    return None


refs = {}


def reference(seed):
    if seed not in refs:
        c = _random_inputs(seed, mps)
        og, op, ost = O.Pts(c["lats"].ravel(), c["lons"].ravel()), O.Pts(c["plat"], c["plon"]), O.Barnes(c["h"])
        ref2, rvar = O.oi_full(og, c["bg"].ravel(), c["bvar"].ravel(), op, c["obs"], c["ratios"], c["pbg"], c["bvp"], ost, c["mp"], c["allow"])
        # (the analysis of optimal_interpolation equals that of optimal_interpolation_full: the background variances only scale the ratios,
        #  which the first form takes as given -- so it has its own oracle call)
        ref = O.oi(og, c["bg"].ravel(), op, c["obs"], c["ratios"], c["pbg"], ost, c["mp"], c["allow"])
        shp = (c["Y"], c["X"])
        refs[seed] = (c, ref.reshape(shp), ref2.reshape(shp), rvar.reshape(shp))
        if spatial:   # scales +-25 % over the domain, looked up at the first point of corr(p1, p2)
            hf = (c["h"] * (1 + 0.25 * np.sin(9 * c["lats"]) * np.cos(7 * c["lons"]))).astype(np.float32)
            Rf = np.array([O.structure_localization("Barnes", h, 0.0013) for h in hf.ravel()], np.float32)
            oi_ = O.nearest_indices(og, op)
            z_g, z_p = np.zeros(hf.size, np.float32), np.zeros(c["S"], np.float32)
            refsp, _ = O.oi_full_generic(og, c["bg"].ravel(), np.ones(hf.size, np.float32), op, c["obs"], c["ratios"], c["pbg"], np.ones(c["S"], np.float32),
                                         O.Struct("Barnes", c["h"]), c["mp"], c["allow"], [hf.ravel(), z_g, z_g, Rf], [hf.ravel()[oi_], z_p, z_p, Rf[oi_]])
            refs[seed] += (hf, refsp.reshape(shp))
    return refs[seed]


failures = []
handles = {}
seen = set()


def record(seed, what, detail, c, arrays, stats):
    failures.append((seed, what, detail))
    print("FAIL seed %d [%s]: %s; stats %s; Y=%d X=%d S=%d h=%g mp=%d allow=%d" % (seed, what, detail, stats, c["Y"], c["X"], c["S"], c["h"], c["mp"], c["allow"]), flush=True)
    os.makedirs(dump, exist_ok=True)
    np.savez(os.path.join(dump, "fail_%d_%s_%d.npz" % (seed, what.replace(" ", "_"), len(failures))), **arrays,
             **{k: v for k, v in c.items() if isinstance(v, np.ndarray)})


def one(seed):
    c, ref, ref2, rvar = reference(seed)[:4]
    if reuse:
        if seed not in handles:
            handles[seed] = (gridpp.Grid(c["lats"], c["lons"]), gridpp.Points(c["plat"], c["plon"]))
        grid, points = handles[seed]
    else:
        grid, points = gridpp.Grid(c["lats"], c["lons"]), gridpp.Points(c["plat"], c["plon"])
    st = gridpp.BarnesStructure(c["h"])
    hostile()
    out = gridpp.optimal_interpolation(grid, c["bg"], points, c["obs"], c["ratios"], c["pbg"], st, c["mp"], c["allow"])
    s1 = gridpp.oi_last_stats()
    d = compare(out, ref)
    if d:
        record(seed, "analysis", d, c, dict(out=out, ref=ref), s1)
    if not s1["union_kernel_ms"] > 0 and not (reuse and seed in seen):
        # (with the handles kept, a geometry whose first pass declined more than half of its tiles goes to k_oi alone from its second call on)
        record(seed, "first pass did not run", "", c, dict(out=out, ref=ref), s1)
    seen.add(seed)
    hostile()
    out2, var = gridpp.optimal_interpolation_full(grid, c["bg"], c["bvar"], points, c["obs"], c["ratios"], c["pbg"], c["bvp"], st, c["mp"], c["allow"])
    s2 = gridpp.oi_last_stats()
    d = compare(out2, ref2)
    if d:
        record(seed, "analysis (full)", d, c, dict(out=out2, ref=ref2), s2)
    d = compare(var, rvar)
    if d:
        record(seed, "variance", d, c, dict(out=var, ref=rvar), s2)
    # the same call again (now the geometry remembers the last call): bit-identical
    hostile()
    out3, var3 = gridpp.optimal_interpolation_full(grid, c["bg"], c["bvar"], points, c["obs"], c["ratios"], c["pbg"], c["bvp"], st, c["mp"], c["allow"])
    s3 = gridpp.oi_last_stats()
    if reuse:   # the deferred form on device-resident copies of the same inputs: the bits of the blocking call
        import torch
        dev = [torch.from_numpy(np.ascontiguousarray(c[k], dtype=np.float32)).cuda() for k in ("bg", "obs", "ratios", "pbg")]
        hostile()
        pend = gridpp.optimal_interpolation_async(grid, dev[0], points, dev[1], dev[2], dev[3], st, c["mp"], c["allow"])
        pend2 = gridpp.optimal_interpolation_async(grid, dev[0], points, dev[1], dev[2], dev[3], st, c["mp"], c["allow"])
        oa, ob = pend.wait().cpu().numpy(), pend2.wait().cpu().numpy()
        if not (np.array_equal(oa, out, equal_nan=True) and np.array_equal(ob, out, equal_nan=True)):
            record(seed, "deferred call differs", "%d / %d values" % (int((~((oa == out) | (np.isnan(oa) & np.isnan(out)))).sum()), int((~((ob == out) | (np.isnan(ob) & np.isnan(out)))).sum())),
                   c, dict(out=out, deferred=oa, deferred2=ob, ref=ref), gridpp.oi_last_stats())
    if spatial:
        hf, refsp = reference(seed)[4:]
        z = np.zeros_like(hf)
        sst = gridpp.BarnesStructure(grid, hf, z, z, 0.0013)
        hostile()
        o1 = gridpp.optimal_interpolation(grid, c["bg"], points, c["obs"], c["ratios"], c["pbg"], sst, c["mp"], c["allow"])
        ss = gridpp.oi_last_stats()
        d = compare(o1, refsp)
        if d:
            record(seed, "analysis (spatially varying)", d, c, dict(out=o1, ref=refsp, hf=hf), ss)
        if not ss["union_kernel_ms"] > 0:
            record(seed, "tile path did not run (spatially varying)", "", c, dict(out=o1, ref=refsp, hf=hf), ss)
        hostile()
        o2 = gridpp.optimal_interpolation(grid, c["bg"], points, c["obs"], c["ratios"], c["pbg"], sst, c["mp"], c["allow"])
        if not np.array_equal(o1, o2, equal_nan=True):
            record(seed, "repeat differs (spatially varying)", "%d values" % int((~((o1 == o2) | (np.isnan(o1) & np.isnan(o2)))).sum()), c, dict(out=o1, out_again=o2, ref=refsp, hf=hf), ss)
    if not (np.array_equal(out2, out3, equal_nan=True) and np.array_equal(var, var3, equal_nan=True)):
        nd = int((~((out2 == out3) | (np.isnan(out2) & np.isnan(out3)))).sum()) + int((~((var == var3) | (np.isnan(var) & np.isnan(var3)))).sum())
        record(seed, "repeat differs", "%d values; second call: %s" % (nd, s3), c, dict(out=out2, out_again=out3, var=var, var_again=var3, ref=ref2), s2)


def unrelated(rng):
    """a larger call of another shape in between (other workspaces sizes, other kernels resident last)"""
    Y, X, S = int(rng.integers(150, 400)), int(rng.integers(150, 400)), int(rng.integers(500, 3000))
    lats, lons = np.meshgrid(np.linspace(60, 61, Y), np.linspace(10, 12, X), indexing="ij")
    g, p = gridpp.Grid(lats, lons), gridpp.Points(60 + rng.random(S), 10 + 2 * rng.random(S))
    f = rng.normal(0, 1, (Y, X)).astype(np.float32)
    v = rng.normal(0, 1, S).astype(np.float32)
    gridpp.optimal_interpolation(g, f, p, v, np.abs(v) + 0.1, v, gridpp.BarnesStructure(float(rng.choice([4000.0, 15000.0]))), int(rng.choice([5, 30, 45, 62, 0][:4])))


t0 = time.time()
for seed in range(lo, hi):
    reference(seed)
print("oracle answers for seeds %d..%d (max_points %s) in %.0f s" % (lo, hi, mps, time.time() - t0), flush=True)
for rep in range(repeats):
    rng = np.random.default_rng(12345 + rep)
    order = np.arange(lo, hi) if rep == 0 else rng.permutation(np.arange(lo, hi))
    t1, nf = time.time(), len(failures)
    for k, seed in enumerate(order):
        if rep > 0 and k % 37 == 5:
            unrelated(rng)
        one(int(seed))
    print("pass %d (%s order%s): %d failures in %.0f s" % (rep, "sequential" if rep == 0 else "shuffled", ", poisoned" if poison else "", len(failures) - nf, time.time() - t1), flush=True)
print("FAILURES %d" % len(failures))
for f in failures[:20]:
    print(f)
sys.exit(1 if failures else 0)
