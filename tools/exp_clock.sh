#!/bin/bash
# shader clock and power while one dense-layer product runs back to back: tools/exp_clock.sh nt|nn|tn [ARCN_GEMM_SPLIT]
KIND=${1:-nt}; SPLIT=${2:-1}
ARCN_GEMM_SPLIT=$SPLIT python - "$KIND" <<'PY' &
import sys, time, torch
sys.path.insert(0, '.')
from arcnerf_amd.ops import functional as F
kind = sys.argv[1]
S, K, N = 1 << 20, 256, 256
x = torch.randn(S, K, device='cuda'); w = torch.randn(N, K, device='cuda'); dy = torch.randn(S, N, device='cuda'); b = torch.randn(N, device='cuda')
t0 = time.time()
while time.time() - t0 < 8:
    for _ in range(50):
        if kind == 'nt': F.gemm_nt(x, w, b)
        elif kind == 'nn': F.gemm_nn(dy, w)
        elif kind == 'mm': torch.mm(x, w.t())
        else: F.gemm_tn(dy, x)
    torch.cuda.synchronize()
PY
PID=$!
sleep 4
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | head -6; sleep 1; done
wait $PID
