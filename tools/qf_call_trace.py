"""Where one neighbourhood_quantile_fast call of config 4 (4000 x 4000 x 100, halfwidth 15, 11 thresholds, device-resident cube) spends its time:
a kernel + memory-copy trace of a few steady-state calls next to the wall clock of the call.

    python tools/qf_call_trace.py run [ny]      # the workload (what rocprofv3 traces); prints ms per call
    python tools/qf_call_trace.py parse DIR     # the timeline of one steady-state call from the trace under DIR
on the GPU box: cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/qt -- python $REPO/tools/qf_call_trace.py run;
                python $REPO/tools/qf_call_trace.py parse /tmp/qt"""
import csv, glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import torch
    import gridpp_amd as gridpp
    from tools.bench_cases import c4_cube
    ny = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    cube = c4_cube(ny, ny, 100)
    thr = torch.linspace(0, 10, 11, device="cuda")
    for _ in range(3):
        gridpp.neighbourhood_quantile_fast(cube, 0.5, 15, thr)
    torch.cuda.synchronize()
    K = 10
    t0 = time.perf_counter()
    for _ in range(K):
        gridpp.neighbourhood_quantile_fast(cube, 0.5, 15, thr)
    torch.cuda.synchronize()
    print("quantile_fast %d x %d x 100, hw 15, T 11: %.3f ms per call (wall, %d calls)" % (ny, ny, (time.perf_counter() - t0) / K * 1e3, K))
    t0 = time.perf_counter()
    for _ in range(K):
        gridpp.neighbourhood(cube, 15, gridpp.Mean)
    torch.cuda.synchronize()
    print("neighbourhood Mean: %.3f ms per call (wall, %d calls)" % ((time.perf_counter() - t0) / K * 1e3, K))
else:
    ev = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "")[:60]))
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy %s" % r.get("Direction", r.get("Name", ""))))
    ev.sort()
    starts = [i for i, e in enumerate(ev) if "k_qf_count" in e[2]]
    i0, i1 = starts[8], starts[9]          # the ninth call of the run: steady state
    # the call begins with its first event behind the previous call's box pass
    j = i0
    while j > 0 and "k_qf_box" not in ev[j - 1][2]:
        j -= 1
    t0 = ev[j][0]
    print("# one steady-state call; microseconds relative to its first GPU event")
    print("# %-60s %10s %10s %10s" % ("event", "start", "end", "duration"))
    k = j
    while k < len(ev) and (k <= i0 or "k_qf_count" not in ev[k][2]):
        s, e, n = ev[k]
        print("%-62s %10.1f %10.1f %10.1f" % (n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
        if "k_qf_count" in n and k > i0:
            break
        k += 1
        if k < len(ev) and "k_qf_lut" in ev[k][2] and k > i0:
            print("# next call's first event at %.1f" % ((ev[k][0] - t0) / 1e3))
            break
