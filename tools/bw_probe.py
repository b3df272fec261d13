import torch, time
g = torch.Generator(device="cuda").manual_seed(1003)
cube = torch.rand((4000, 4000, 100), generator=g, device="cuda") * 10
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts=[]
    for _ in range(n):
        t0=time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    return min(ts)
b = cube.numel()*4
for name, fn in [("sum(dim=2)", lambda: cube.sum(dim=2)), ("sum all", lambda: cube.sum()), ("clone", lambda: cube.clone())]:
    dt = t(fn)
    print(name, "%.3f ms" % (dt*1e3), "%.2f TB/s read" % (b/dt/1e12))
