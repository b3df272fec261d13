import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import bench_paths as bp
bp.oi_case("C3 plain", 4000, 4000, 10000, 30, 1002)
