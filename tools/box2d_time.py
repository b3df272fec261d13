"""neighbourhood(Mean) of one 4000 x 4000 plane, halfwidth 15 (the box pass of config 4 alone): ms per call; `tools/kstats.sh python tools/box2d_time.py`
for the kernel times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gridpp_amd as gridpp
hw = int(sys.argv[1]) if len(sys.argv) > 1 else 15
g = torch.Generator(device="cuda").manual_seed(5)
plane = torch.rand((4000, 4000), generator=g, device="cuda") * 10
for _ in range(3):
    gridpp.neighbourhood(plane, hw, gridpp.Mean)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    gridpp.neighbourhood(plane, hw, gridpp.Mean)
torch.cuda.synchronize()
print("neighbourhood(4000 x 4000, halfwidth %d, Mean): %.3f ms per call" % (hw, (time.perf_counter() - t0) / 20 * 1e3))
