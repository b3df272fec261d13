"""Evidence that the deferred calls overlap as DESIGN 11.4 says: a kernel trace (rocprofv3 --kernel-trace) of one rank's share of the headline at
N = 8 (500 rows) run as a stream of analyses with one ahead; prints, for a few consecutive steady-state calls, when each kernel started and
ended relative to the first pass of the first of them.

    python tools/overlap_trace.py run          # the workload (what rocprofv3 traces)
    python tools/overlap_trace.py parse DIR    # the timeline from the trace under DIR
usage on the GPU box: cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/ot -- python $REPO/tools/overlap_trace.py run; python $REPO/tools/overlap_trace.py parse /tmp/ot"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import numpy as np, torch
    import gridpp_amd as gridpp
    from gridpp_amd.dist import AnalysisPipeline
    from bench import make_workload
    lats, lons, bg, plat, plon, obs, ratios, pbg = make_workload(4000, 4000, 10000, 1002, 0, 500)
    grid, points, st = gridpp.Grid(lats, lons), gridpp.Points(plat, plon), gridpp.BarnesStructure(10000)
    d = [torch.from_numpy(a).cuda() for a in (bg, obs, ratios, pbg)]
    pipe = AnalysisPipeline(1)
    for _ in range(12):
        pipe.push(gridpp.optimal_interpolation_async(grid, d[0], points, d[1], d[2], d[3], st, 30))
    pipe.drain()
    torch.cuda.synchronize()
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_oi" in r["Kernel_Name"] or "k_pack_obs" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    firsts = [r for r in rows if "k_oi_union<true, false" in r["Kernel_Name"]]
    t0 = int(firsts[6]["Start_Timestamp"])                      # the seventh first pass: steady state
    t1 = int(firsts[9]["End_Timestamp"])
    print("# one rank's share at N = 8 (500 rows x 4000 columns), deferred calls with one analysis ahead; microseconds relative to the start of a first pass")
    print("# %-44s %10s %10s %10s   queue / stream" % ("kernel", "start", "end", "duration"))
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t0 - 700000 <= s <= t1:
            print("%-46s %10.1f %10.1f %10.1f   %s" % (r["Kernel_Name"].replace("void ", "")[:46], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", r.get("Stream_Id", ""))))
