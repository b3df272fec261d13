"""The 'round-N numbers' table of DESIGN.md section 6 from a bench line (profiles/rNN_bench_n1.json): markdown rows on stdout."""
import json, sys
d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "profiles/r04_bench_n1.json"))
k, cb = d["kernel"], d["cpu_baseline"]
print("| case | time | throughput | algorithmic GB/s (of 8 TB/s) | note |")
print("|---|---|---|---|---|")
print("| C3 OI 4000², 10 k obs, mp 30 (headline) | %.2f ms/step | %.2f Gcells/s | %.1f (%.1f %%) | `k_oi_union` %.2f ms, all OI kernels %.2f ms; VALU issue %.0f–%.0f %%; host-inclusive %.1f ms; CPU port: %.1f kcells/s on 1 thread, %.0f kcells/s on %d threads (cgroup quota) |"
      % (d["ms_per_step"], d["value"] / 1e9, d["roofline"]["achieved"], 100 * d["roofline"]["frac"], k["avg_ms"], k["all_oi_kernels_ms"],
         100 * d["roofline_compute"]["frac_bounds"][0], 100 * d["roofline_compute"]["frac_bounds"][1], d["host_inclusive"]["ms_per_step"],
         cb["one_thread_value"] / 1e3, cb["value"] / 1e3, cb["cores"]))
for c in d["other_configs"]:
    extra = ""
    if "declined_tiles" in c:
        extra = "%d tiles declined by the first pass, %d factorisations" % (c["declined_tiles"], c["solves"])
    if "reference_formulation_TFLOPs_equivalent" in c:
        extra = "%s" % ", ".join("%s %s" % (kk, (round(v, 2) if isinstance(v, float) else v)) for kk, v in c.items() if "TFLOP" in kk)
    print("| %s | %.3g ms | %.3g Mcells/s | %.0f (%.1f %%) | %s |" % (c["case"], c["ms"], c["Mcells/s"], c["GB/s_algorithmic"], 100 * c["frac_hbm"], extra))
