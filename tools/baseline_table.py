"""The results table of BASELINE.md section 4 from a bench line (profiles/rNN_bench_n1.json): markdown rows on stdout."""
import json, sys
d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "profiles/r04_bench_n1.json"))
cb, k = d["cpu_baseline"], d["kernel"]
oc = {o["case"].split(",")[0].split(" hw")[0] if o["case"].startswith("C4") else o["case"]: o for o in d["other_configs"] if "case" in o and "ms" in o}
def find(prefix):
    return next(o for o in d["other_configs"] if o.get("case", "").startswith(prefix))
def cpu(o):
    c = o.get("cpu_baseline")
    if not c:
        return "—"
    if "all_threads_value" in c:
        return "%.0f cells/s (1 thread: the reference's loop is serial); %.0f cells/s on %d threads" % (c["value"], c["all_threads_value"], c["all_threads_cores"])
    return "%.1f Mcells/s (%d threads), %.1f Mcells/s (1)" % (c["value"] / 1e6, c["cores"], c["one_thread_value"] / 1e6)
print("| config | CPU restatement (cores) | 1 GPU | HBM GB/s (algorithmic) | % HBM roofline | GPU / CPU |")
print("|---|---|---|---|---|---|")
c1, c2, c3s, c3n, c4m, c4q, c5 = find("C1"), find("C2"), find("C3 OI 4000x4000, 10k obs, mp=30, smooth"), find("C3 OI 4000x4000, 10k obs, mp=30, white"), find("C4 neighbourhood Mean"), find("C4 quantile_fast"), find("C5")
print("| C1 200², 10 obs, mp=10 | — | %.0f Mcells/s (%.3f ms per call: launch-bound) | %.0f | %.1f %% | — |" % (c1["Mcells/s"], c1["ms"], c1["GB/s_algorithmic"], 100 * c1["frac_hbm"]))
print("| C2 1000², 1k obs, mp=20 | — | %.0f Mcells/s (%.3f ms) | %.0f | %.1f %% | — |" % (c2["Mcells/s"], c2["ms"], c2["GB/s_algorithmic"], 100 * c2["frac_hbm"]))
print("| **C3 4000², 10k obs, mp=30 (headline)** | %.1f kcells/s (1), %.1f kcells/s (%d) | **%.0f Mcells/s (%.2f ms per step**; `k_oi_union` %.2f ms, all kernels %.2f ms; from numpy buffers %.1f ms) | %.0f | %.1f %% | %.0f × |"
      % (cb["one_thread_value"] / 1e3, cb["value"] / 1e3, cb["cores"], d["value"] / 1e6, d["ms_per_step"], k["avg_ms"], k["all_oi_kernels_ms"], d["host_inclusive"]["ms_per_step"],
         d["roofline"]["achieved"], 100 * d["roofline"]["frac"], d["speedup_vs_cpu_baseline"]))
print("| C3, smooth terrain (elev + laf, v = 200 m, w = 0.5) | — | %.0f Mcells/s (%.2f ms) | %.0f | %.1f %% | — |" % (c3s["Mcells/s"], c3s["ms"], c3s["GB/s_algorithmic"], 100 * c3s["frac_hbm"]))
print("| C3, white-noise terrain | — | %.0f Mcells/s (%.1f ms) | %.1f | %.2f %% | — |" % (c3n["Mcells/s"], c3n["ms"], c3n["GB/s_algorithmic"], 100 * c3n["frac_hbm"]))
for o, nm in ((c4m, "C4 4000²×100, hw=15, Mean"), (c4q, "C4 4000²×100, hw=15, quantile_fast (11 thresholds)")):
    print("| %s | %s | %.0f Mcells/s (%.2f ms) | %.0f | %.0f %% | %.0f × |" % (nm, cpu(o), o["Mcells/s"], o["ms"], o["GB/s_algorithmic"], 100 * o["frac_hbm"], o.get("speedup_vs_cpu_baseline", float("nan"))))
print("| C5 EnSI 2500²×50, 5k obs, mp=30 | %s | %.1f Mcells/s (%.1f ms default mode; %.1f ms with the sweeps run to convergence = %.1f Mcells/s; %.1f TFLOP/s executed FP64 = %.0f %% of peak) | %.1f | %.1f %% | %.0f × (1 thread); see the line below |"
      % (cpu(c5), c5["Mcells/s"], c5["ms"], c5.get("ms_converged", float("nan")), c5.get("Mcells/s_converged", float("nan")), c5.get("fp64_TFLOPs_executed", float("nan")),
         100 * c5.get("frac_fp64_peak_executed", float("nan")), c5["GB/s_algorithmic"], 100 * c5["frac_hbm"], c5.get("speedup_vs_cpu_baseline", float("nan"))))
# (round 6: the C5 ratio against BOTH thread counts -- the 1-thread figure is the reference's own serial loop, the all-threads figure what its commented-out pragma would give)
cb5 = c5.get("cpu_baseline") or {}
if "all_threads_value" in cb5:
    print("C5 GPU / CPU: %.0f x against 1 thread (%.0f cells/s), %.0f x against %d threads (%.0f cells/s)" % (
        c5["Mcells/s"] * 1e6 / cb5["value"], cb5["value"], c5["Mcells/s"] * 1e6 / cb5["all_threads_value"], cb5["all_threads_cores"], cb5["all_threads_value"]))
