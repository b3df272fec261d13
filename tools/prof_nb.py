import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gridpp_amd as gridpp
g = torch.Generator(device="cuda").manual_seed(1003)
cube = torch.rand((4000, 4000, 100), generator=g, device="cuda") * 10
thr = torch.linspace(0, 10, 11, device="cuda")
for _ in range(3):
    gridpp.neighbourhood(cube, 15, gridpp.Mean)
    gridpp.neighbourhood_quantile_fast(cube, 0.5, 15, thr)
torch.cuda.synchronize()
