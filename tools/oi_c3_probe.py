"""One C3 call (4000 x 4000, 10 000 obs, max_points 30) on flat ground, smooth terrain or white-noise terrain, with the statistics of the call:
    python tools/oi_c3_probe.py plain|smooth|noise
With GPP_LIB=gridpp_amd/lib/var_<name>.so (tools/variant.sh <name> oi -DGPP_UNION_PROFILE -DGPP_UNION_STATS -DGPP_TIMING_SWITCHES) the library prints
the per-phase clocks of k_oi_union and its scan statistics (DESIGN.md 4.1 "Round 4: the scan on terrain")."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_paths as bp
import gridpp_amd as gridpp
which = sys.argv[1] if len(sys.argv) > 1 else "plain"
bp.oi_case("C3 " + which, 4000, 4000, 10000, 30, 1002, elev={"plain": False, "smooth": True, "noise": "noise"}[which])
print(gridpp.oi_last_stats())
