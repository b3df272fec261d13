"""neighbourhood(Max) of one 4000 x 4000 plane (device resident), halfwidth 15 by default: ms per call (k_minmax_march against the two passes of k_minmax_pass with GPP_BOX_TWO_PASS=1)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gridpp_amd as gridpp
hw = int(sys.argv[1]) if len(sys.argv) > 1 else 15
g = torch.Generator(device="cuda").manual_seed(5)
plane = torch.rand((4000, 4000), generator=g, device="cuda") * 10
for _ in range(3):
    gridpp.neighbourhood(plane, hw, gridpp.Max)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    gridpp.neighbourhood(plane, hw, gridpp.Max)
torch.cuda.synchronize()
print("neighbourhood(4000 x 4000, halfwidth %d, Max): %.3f ms per call" % (hw, (time.perf_counter() - t0) / 20 * 1e3))
