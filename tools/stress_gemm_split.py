"""random shapes: split products (and the bit-mask / column-sum options) against float64; python tools/stress_gemm_split.py [n=40]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from arcnerf_amd.ops import functional as F
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    S = int(rng.choice([1, 7, 8, 31, 33, 127, 129, 1000, 4097, 20011, 131072 + 5]))
    K = 4 * int(rng.integers(17, 100))
    Nn = 4 * int(rng.integers(17, 100))
    print('shape', S, K, Nn, flush=True)
    x = torch.randn(S, K, device='cuda'); w = torch.randn(Nn, K, device='cuda') / K ** 0.5; b = torch.randn(Nn, device='cuda'); dy = torch.randn(S, Nn, device='cuda')
    y, bits = F.gemm_nt(x, w, b, act='relu', want_bits=True)
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    e = [((y.double() - ref).abs().max() / max(1.0, ref.abs().max().item())).item()]
    m = (y > 0)
    refx = (dy.double() * m) @ w.double()
    e.append(((F.gemm_nn(dy, w, mask_bits=bits).double() - refx).abs().max() / max(1e-6, refx.abs().max().item())).item())
    e.append(((F.gemm_nn(dy, w, mask=y).double() - refx).abs().max() / max(1e-6, refx.abs().max().item())).item())
    refw = (dy.double() * m).t() @ x.double(); refb = (dy.double() * m).sum(0)
    dw, db = F.gemm_tn(dy, x, mask_bits=bits, want_colsum=True)
    sc = max(1.0, (S / 1000.0) ** 0.5)
    e.append(((dw.double() - refw).abs().max() / max(1e-6, refw.abs().max().item())).item() / sc)
    e.append(((db.double() - refb).abs().max() / max(1e-6, refb.abs().max().item())).item() / sc)
    dw2 = F.gemm_tn(dy, x)
    refw2 = dy.double().t() @ x.double()
    e.append(((dw2.double() - refw2).abs().max() / max(1e-6, refw2.abs().max().item())).item() / sc)
    torch.cuda.synchronize()
    ok = max(e) < 5e-6
    bad += not ok
    print(S, K, Nn, ' '.join('%.1e' % v for v in e), '' if ok else '  <-- BAD')
print('bad', bad)
