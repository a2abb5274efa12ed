"""experiment: where does the fused MLP time go"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arcnerf_amd import _native as N
from arcnerf_amd.ops import functional as F
dev = torch.device('cuda:0')
def timeit(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
n = 260608
for name, dims, ao in (('geo', [32, 64, 16], None), ('rad', [32, 64, 64, 3], 'sigmoid'), ('one', [32, 64], None), ('tiny', [16, 16], None)):
    desc = N.make_mlp_desc(dims, 'relu', ao)
    nw = sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
    w = torch.randn(nw, device=dev) * 0.1
    x = torch.randn(n, dims[0], device=dev)
    out = torch.empty(n, dims[-1], device=dev)
    acts = torch.empty(max(1, F.mlp_acts_floats(desc, n)), device=dev)
    t0 = timeit(lambda: F.mlp_fwd(x, w, None, desc, out=out))
    t1 = timeit(lambda: F.mlp_fwd(x, w, None, desc, save_acts=True, out=out, acts=acts))
    dout = torch.randn(n, dims[-1], device=dev)
    dw = torch.zeros_like(w); scr = torch.empty(F.mlp_scratch_floats(desc, n), device=dev)
    t2 = timeit(lambda: F.mlp_bwd(x, w, None, desc, out, acts, dout, dweights=dw, scratch=scr))
    import ctypes as C
    dxb = torch.empty_like(x)
    def dx_only():
        N.check(N.lib().arcn_mlp_bwd(x.data_ptr(), w.data_ptr(), None, C.addressof(desc), out.data_ptr(), acts.data_ptr(), dout.data_ptr(),
                                     dxb.data_ptr(), None, None, scr.data_ptr(), n, n, None, N.stream()))
    t3 = timeit(dx_only)
    print('%-5s fwd %.1f us  fwd+acts %.1f us  bwd(dx+dw) %.1f us  dx only %.1f us -> dw %.1f us' % (name, t0, t1, t2, t3, t2 - t3))
# plain copy bandwidth reference
a = torch.empty(n * 64, device=dev); b = torch.empty_like(a)
print('copy 67MB: %.1f us' % timeit(lambda: b.copy_(a)))
