"""Timeline of ONE headline optimal_interpolation call from numpy arrays (the banded host path): kernels and memory copies relative to the call's first GPU event.

    python tools/host_path_trace.py run         # the workload (what rocprofv3 traces); prints ms per call
    python tools/host_path_trace.py parse DIR   # the timeline of one steady-state call
on the GPU box: cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ht -- python $REPO/tools/host_path_trace.py run;
                python $REPO/tools/host_path_trace.py parse /tmp/ht"""
import csv, glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import numpy as np
    import gridpp_amd as gridpp
    from tools.bench_cases import make_workload
    lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(4000, 4000, 10000, 1002, 0, 4000)
    grid, points, st = gridpp.Grid(lats, lons), gridpp.Points(plat, plon), gridpp.BarnesStructure(10000)
    for _ in range(3):
        gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, 30)
    t0 = time.perf_counter()
    for _ in range(8):
        gridpp.optimal_interpolation(grid, bg, points, obs, ratios, pbg, st, 30)
    print("%.3f ms per call" % ((time.perf_counter() - t0) / 8 * 1e3))
else:
    ev = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "")[:50] + " grid %s" % r.get("Grid_Size", "")))
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy %s" % r.get("Direction", r.get("Name", ""))))
    ev.sort()
    packs = [i for i, e in enumerate(ev) if "k_pack_obs" in e[2]]
    i0, i1 = packs[-3], packs[-2]          # one steady-state call: from its k_pack_obs (behind its small uploads) to the next one's
    j = i0
    while j > 0 and ev[j - 1][2].startswith("memcpy") and ev[i0][0] - ev[j - 1][0] < 300000:
        j -= 1
    t0 = ev[j][0]
    print("# one steady-state call from numpy arrays; microseconds relative to its first GPU event")
    print("# %-72s %10s %10s %10s" % ("event", "start", "end", "duration"))
    for s, e, n in ev[j:i1]:
        if n.startswith("memcpy") and e - s < 3000 and not (j <= ev.index((s, e, n)) <= i0):
            pass
        print("%-74s %10.1f %10.1f %10.1f" % (n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
