import os, sys, torch, time
sys.path.insert(0, os.getcwd())
from arcnerf_amd.ops import functional as F
dev = torch.device('cuda:0')
z = torch.rand(131072, 256, device=dev) * 0.05
g = torch.randn(131072, 256, device=dev)
h = torch.randn(131072, 256, device=dev)
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print('softplus_grad  us', t(lambda: F.softplus_grad(z, g, 100.0, True)))
print('softplus_grad2 us', t(lambda: F.softplus_grad2(z, g, h, 100.0, from_y=True)))
print('softplus_grad_sum us', t(lambda: F.softplus_grad_sum(z, g, h, 100.0, True)))
