"""one split / exact dense-layer product repeated (for counter passes): python tools/exp_gemm_one.py nt|nn|tn [S K N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arcnerf_amd.ops import functional as F
kind = sys.argv[1] if len(sys.argv) > 1 else 'nt'
S, K, N = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (1 << 20, 256, 256)
x = torch.randn(S, K, device='cuda'); w = torch.randn(N, K, device='cuda'); dy = torch.randn(S, N, device='cuda'); b = torch.randn(N, device='cuda')
for _ in range(5):
    if kind == 'nt':
        F.gemm_nt(x, w, b)
    elif kind == 'nn':
        F.gemm_nn(dy, w)
    else:
        F.gemm_tn(dy, x)
torch.cuda.synchronize()
