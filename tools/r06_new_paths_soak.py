"""Randomised soak of the code paths round 6 added, beyond the seeds of the test suite:
  sp      k_oi_union_sp (spatially varying Barnes on the tile path) against the oracle and against the pivoted LU per selection
  c48     k_oi_union<., ., 48> (max_points 33..48) against the oracle
  pair    k_oi_union_pair (behind its switch) against the oracle
  bands   the banded host path of optimal_interpolation (numpy in / numpy out, results written in place) against the unbanded one, bit for bit,
          on random shapes of >= 2^20 cells, float32 / float64, with / without variance
  nbh     the banded host path of neighbourhood against the unbanded one, bit for bit
usage: python tools/r06_new_paths_soak.py SECONDS [parts...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gridpp_amd as gridpp
from tests import test_gpu_oi_union_stress as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
parts = sys.argv[2:] or ["sp", "c48", "pair", "bands", "nbh"]
per = budget / len(parts)


class MP:   # what the tests use of pytest's monkeypatch, on the library's own hook
    def setenv(self, k, v): gridpp.set_path_override(k, v)
    def delenv(self, k): gridpp.set_path_override(k, None)


def bits_equal(a, b):
    return a.shape == b.shape and (np.asarray(a).view(np.uint32) == np.asarray(b).view(np.uint32)).all()


fails = 0
for part in parts:
    t0, n = time.time(), 0
    seed = 100000
    while time.time() - t0 < per:
        seed += 1
        try:
            if part == "sp":
                T.test_random_configurations_spatially_varying_barnes_on_the_tile_path(seed, MP())
                gridpp.set_path_override("GPP_OI_NO_SP_UNION", None)
            elif part == "c48":
                T._random_configuration(seed, [33, 36, 40, 44, 46, 47, 48], cressman=seed % 7 == 0)
            elif part == "pair":
                gridpp.set_path_override("GPP_OI_PAIR_TILES", "1")
                T._random_configuration(seed, [1, 2, 7, 20, 30, 32])
                gridpp.set_path_override("GPP_OI_PAIR_TILES", None)
            elif part == "bands":
                rng = np.random.default_rng(seed)
                Y = int(rng.integers(520, 2200)); X = int(max(1 << 20, rng.integers(1 << 20, 3 << 20)) // Y + 1)
                S = int(rng.choice([300, 2000, 6000]))
                lats, lons = np.meshgrid(np.linspace(60, 61.5, Y), np.linspace(10, 13, X), indexing="ij")
                plat, plon = rng.uniform(60, 61.5, S), rng.uniform(10, 13, S)
                dt = np.float64 if seed % 3 == 0 else np.float32
                bg = rng.normal(0, 1, (Y, X)).astype(dt); bvar = rng.uniform(0.5, 2, (Y, X)).astype(dt)
                obs, pbg = rng.normal(0, 1, S).astype(dt), rng.normal(0, 1, S).astype(dt)
                ovar, bvp = rng.uniform(0.1, 1, S).astype(dt), rng.uniform(0.5, 2, S).astype(dt)
                points, st = gridpp.Points(plat, plon), gridpp.BarnesStructure(float(rng.choice([5000.0, 12000.0])))
                mp_ = int(rng.choice([5, 20, 32]))
                full = seed % 2 == 0
                # The SAME sequence of calls on two fresh Grid handles, unbanded and banded: the k-th calls must agree bit for bit.  (Not "every call equals the
                # first": a geometry whose first pass declines more than half of its tiles goes to k_oi alone from its second call on, another elimination order --
                # one variance in 1.2 M cells then differs by one float32 ulp between the first and the second call, banded or not.)
                def seq(no_bands):
                    gridpp.set_path_override("GPP_OI_NO_BANDS", "1" if no_bands else None)
                    g = gridpp.Grid(lats, lons)
                    outs = []
                    for _ in range(3):
                        if full:
                            a, v = gridpp.optimal_interpolation_full(g, bg, bvar, points, obs, ovar, pbg, bvp, st, mp_)
                            outs.append((np.array(a), np.array(v)))
                        else:
                            outs.append((np.array(gridpp.optimal_interpolation(g, bg, points, obs, ovar, pbg, st, mp_)), None))
                    return outs
                ref, got = seq(True), seq(False)
                gridpp.set_path_override("GPP_OI_NO_BANDS", None)
                for k, ((ra, rv), (a, v)) in enumerate(zip(ref, got)):
                    assert bits_equal(a, ra) and (rv is None or bits_equal(v, rv)), "banded differs from unbanded in call %d" % k
            elif part == "nbh":
                rng = np.random.default_rng(seed)
                Y = int(rng.integers(300, 3000)); X = int((1 << 20) // Y + rng.integers(1, 600))
                f = rng.uniform(-5, 10, (Y, X)).astype(np.float64 if seed % 3 == 0 else np.float32)
                f[rng.random((Y, X)) < 0.002] = np.nan
                stat, hw = [(gridpp.Mean, int(rng.integers(0, 17))), (gridpp.Sum, int(rng.integers(0, 17))), (gridpp.Count, int(rng.integers(0, 17))), (gridpp.Min, int(rng.integers(0, 33))), (gridpp.Max, int(rng.integers(0, 33)))][seed % 5]
                gridpp.set_path_override("GPP_NBH_NO_BANDS", "1"); ref = np.array(gridpp.neighbourhood(f, hw, stat))
                gridpp.set_path_override("GPP_NBH_NO_BANDS", None)
                assert bits_equal(np.array(gridpp.neighbourhood(f, hw, stat)), ref), "banded neighbourhood differs"
            n += 1
        except AssertionError as e:
            fails += 1
            print("FAIL %s seed %d: %s" % (part, seed, str(e)[:200]), flush=True)
            for k in ("GPP_OI_NO_SP_UNION", "GPP_OI_PAIR_TILES", "GPP_OI_NO_BANDS", "GPP_NBH_NO_BANDS"):
                gridpp.set_path_override(k, None)
    print("%-6s seeds %d..%d: %d configurations in %.0f s" % (part, 100001, seed, n, time.time() - t0), flush=True)
print("FAILURES %d" % fails)
sys.exit(1 if fails else 0)
