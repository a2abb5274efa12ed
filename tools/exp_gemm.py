"""the three dense-layer products of csrc/gemm.hip against torch (hipBLASLt) on the shapes of the wide nets: TFLOP/s of each"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arcnerf_amd.ops import functional as F


def t(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3


shapes = [(1 << 20, 256, 256), (1 << 20, 319, 256), (1 << 20, 63, 256), (1 << 20, 283, 128), (1 << 20, 128, 3), (1 << 18, 32, 64), (1 << 18, 64, 17)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for S, K, N in shapes:
    x = torch.randn(S, K, device='cuda'); w = torch.randn(N, K, device='cuda'); dy = torch.randn(S, N, device='cuda'); b = torch.randn(N, device='cuda')
    fl = 2.0 * S * K * N
    r = []
    for name, mine, ref in (('nt', lambda: F.gemm_nt(x, w, b), lambda: torch.addmm(b, x, w.t())), ('nn', lambda: F.gemm_nn(dy, w), lambda: dy @ w),
                            ('tn', lambda: F.gemm_tn(dy, x), lambda: dy.t() @ x)):
        a, c = t(mine), t(ref)
        r.append('%s %.1f / %.1f TF (%.0f / %.0f us)' % (name, fl / a / 1e12, fl / c / 1e12, a * 1e6, c * 1e6))
    print('S %d K %d N %d: hip / torch  ' % (S, K, N) + ' | '.join(r))
