"""experiment + check: the planned scatter (arcn_hashgrid_bwd_plan + _lm_planned) against the one-pass form on the bench's sample distribution.
Same dtable to summation-order noise; run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from arcnerf_amd import _native as N
from arcnerf_amd.ops import functional as F
from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays

dev = torch.device('cuda:0')
cfg = NgpConfig()
fld = NgpField(cfg, device=dev)
pipe = NgpPipeline(fld, max_rays=32768, max_samples=1 << 20)
pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, 0)))
o, d = synthetic_rays(8320, seed=0, device=dev)
pipe.forward(o, d, None, train=True)
S = pipe.cap
n_dev = pipe.n_dev
n = int(n_dev.item())
xyz = pipe.buf['xyz']
L = N.lib()
desc = C.addressof(fld.grid_desc)
g = torch.Generator(device='cpu').manual_seed(3)
dout = (torch.randn(16, S, 2, generator=g) * 1e-3).to(dev)      # level-major (L, S, F)
ws = F.hashgrid_bwd_workspace(fld.grid_desc, S, dev)
pf = int(L.arcn_hashgrid_plan_workspace_floats(desc, S))
print('samples', n, 'capacity', S, 'workspace MB', ws.numel() * 4 / 1e6, 'plan workspace MB', pf * 4 / 1e6)
pws = torch.empty(pf, dtype=torch.float32, device=dev)
st = N.stream()
a = torch.zeros_like(fld.view('table'))
b = torch.zeros_like(a)
N.check(L.arcn_hashgrid_bwd_lm(N.ptr(xyz), N.ptr(dout), S, desc, N.ptr(a), N.ptr(ws), ws.numel(), S, n_dev.data_ptr(), st), 'one-pass')
N.check(L.arcn_hashgrid_bwd_plan(N.ptr(xyz), desc, N.ptr(pws), pf, S, n_dev.data_ptr(), st), 'plan')
N.check(L.arcn_hashgrid_bwd_lm_planned(N.ptr(xyz), N.ptr(dout), S, desc, N.ptr(b), N.ptr(pws), pf, N.ptr(ws), ws.numel(), S, n_dev.data_ptr(), st), 'planned')
torch.cuda.synchronize()
err = (a - b).abs().max().item()
print('max |one-pass - planned| %.3e   max |one-pass| %.3e   touched entries %d / %d' % (err, a.abs().max().item(), int((a != 0).sum()), int((b != 0).sum())))
assert err <= 1e-6 * max(1.0, a.abs().max().item()) + 1e-9, 'planned scatter differs'


def timed(fn, it=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


t_one = timed(lambda: N.check(L.arcn_hashgrid_bwd_lm(N.ptr(xyz), N.ptr(dout), S, desc, N.ptr(a), N.ptr(ws), ws.numel(), S, n_dev.data_ptr(), st), 'one-pass'))
t_plan = timed(lambda: N.check(L.arcn_hashgrid_bwd_plan(N.ptr(xyz), desc, N.ptr(pws), pf, S, n_dev.data_ptr(), st), 'plan'))
t_fill = timed(lambda: N.check(L.arcn_hashgrid_bwd_lm_planned(N.ptr(xyz), N.ptr(dout), S, desc, N.ptr(b), N.ptr(pws), pf, N.ptr(ws), ws.numel(), S, n_dev.data_ptr(), st), 'planned'))
print('one-pass scatter %.1f us | plan %.1f us (off the critical path) | fill + owners %.1f us' % (t_one, t_plan, t_fill))
