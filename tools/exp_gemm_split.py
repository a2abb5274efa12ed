"""split-bf16 forms of csrc/gemm.hip: error against float64 next to the exact-f32 MFMA kernels' and the library's, then TFLOP/s.
ARCN_GEMM_SPLIT=0/1 python tools/exp_gemm_split.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arcnerf_amd.ops import functional as F
from exp_gemm import t

torch.manual_seed(0)
print('split', F._GEMM_SPLIT)
for S, K, N in [(4096 + 37, 64, 128), (4096 + 37, 256, 256), (4096 + 37, 284, 260), (4096 + 37, 320, 256), (130, 32, 68)]:
    x = torch.randn(S, K, device='cuda') * torch.rand(S, 1, device='cuda') * 3
    w = torch.randn(N, K, device='cuda') / K ** 0.5
    b = torch.randn(N, device='cuda')
    dy = torch.randn(S, N, device='cuda')
    m = torch.randn(S, N, device='cuda')
    ref = (x.double() @ w.double().t() + b.double())
    e_nt = ((F.gemm_nt(x, w, b).double() - ref).abs().max() / ref.abs().max()).item()
    e_lib = ((torch.addmm(b, x, w.t()).double() - ref).abs().max() / ref.abs().max()).item()
    ref = (dy.double() * (m > 0)) @ w.double()
    e_nn = ((F.gemm_nn(dy, w, m).double() - ref).abs().max() / ref.abs().max()).item()
    ref = (dy.double() * (m > 0)).t() @ x.double()
    e_tn = ((F.gemm_tn(dy, x, m).double() - ref).abs().max() / ref.abs().max()).item()
    print('S %d K %d N %d: max err / max |ref|: nt %.2e (library %.2e) nn %.2e tn %.2e' % (S, K, N, e_nt, e_lib, e_nn, e_tn))
SB = int(os.environ.get('EXP_S', 1 << 20))
for S, K, N in [(SB, 256, 256), (SB, 320, 256), (SB, 64, 256), (SB, 284, 128), (SB * 3 // 4, 256, 260)]:
    x = torch.randn(S, K, device='cuda'); w = torch.randn(N, K, device='cuda'); dy = torch.randn(S, N, device='cuda'); b = torch.randn(N, device='cuda')
    fl = 2.0 * S * K * N
    r = []
    for name, mine in (('nt', lambda: F.gemm_nt(x, w, b)), ('nn', lambda: F.gemm_nn(dy, w)), ('nn masked', lambda: F.gemm_nn(dy, w, dy)), ('tn', lambda: F.gemm_tn(dy, x)),
                       ('tn masked', lambda: F.gemm_tn(dy, x, dy))):
        a = t(mine)
        r.append('%s %.1f TF (%.0f us)' % (name, fl / a / 1e12, a * 1e6))
    print('S %d K %d N %d: ' % (S, K, N) + ' | '.join(r))
