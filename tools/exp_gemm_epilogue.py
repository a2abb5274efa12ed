"""what the forward product's epilogue costs at the NeRF chunk shape (131072 x 256 x 256): plain, + bias, + ReLU, + bit words; the masked
input-gradient product next to it.  python tools/exp_gemm_epilogue.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arcnerf_amd.ops import functional as F
from exp_gemm import t

S, K, N = 131072, 256, 256
x = torch.randn(S, K, device='cuda'); w = torch.randn(N, K, device='cuda') / 16; b = torch.randn(N, device='cuda'); dy = torch.randn(S, N, device='cuda')
y, bits = F.gemm_nt(x, w, b, act='relu', want_bits=True)
ws = F.split_weights(w, False); wst = F.split_weights(w, True)
out = torch.empty(S, N, device='cuda')
for name, fn in (('nt plain', lambda: F.gemm_nt(x, w, None, ws=ws, out=out)),
                 ('nt + bias', lambda: F.gemm_nt(x, w, b, ws=ws, out=out)),
                 ('nt + bias + relu', lambda: F.gemm_nt(x, w, b, act='relu', ws=ws, out=out)),
                 ('nt + bias + relu + bits', lambda: F.gemm_nt(x, w, b, act='relu', want_bits=True, ws=ws, out=out)),
                 ('nn plain', lambda: F.gemm_nn(dy, w, ws=wst)),
                 ('nn bit mask', lambda: F.gemm_nn(dy, w, mask_bits=bits, ws=wst)),
                 ('tn bit mask + colsum', lambda: F.gemm_tn(dy, x, mask_bits=bits, want_colsum=True)),
                 ('tn plain', lambda: F.gemm_tn(dy, x))):
    a = min(t(fn) for _ in range(3))
    print('%-28s %7.1f us  %6.1f TFLOP/s' % (name, a * 1e6, 2.0 * S * K * N / a / 1e12))
