import sys, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import bench_paths as bp
import gridpp_amd as gridpp
bp.oi_case("C3 smooth", 4000, 4000, 10000, 30, 1002, elev=True)
print(gridpp.oi_last_stats())
