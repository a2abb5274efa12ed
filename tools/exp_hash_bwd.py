"""experiment: per-level cost of the hash-grid scatter on the bench sample distribution"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from arcnerf_amd import _native as N
from arcnerf_amd.ops import functional as F
from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays

dev = torch.device('cuda:0')
cfg = NgpConfig()
fld = NgpField(cfg, device=dev)
pipe = NgpPipeline(fld, max_rays=32768, max_samples=1 << 20)
pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, 0)))
o, d = synthetic_rays(8320, seed=0, device=dev)
rgb, _, _ = pipe.forward(o, d, None, train=True)
n = int(pipe.n_dev.item())
xyz = pipe.buf['xyz'][:n].contiguous()
print('samples', n)

def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

table = fld.view('table')
dt = torch.zeros_like(table)
dout = torch.randn(n, 32, device=dev)
print('full bwd plain  %.3f ms' % timeit(lambda: F.hashgrid_bwd(xyz, table, dout, fld.grid_desc, dtable=dt)))
ws = F.hashgrid_bwd_workspace(fld.grid_desc, n, dev)
print('full bwd xcd    %.3f ms' % timeit(lambda: F.hashgrid_bwd(xyz, table, dout, fld.grid_desc, dtable=dt, workspace=ws)))
print('full fwd        %.3f ms' % timeit(lambda: F.hashgrid_fwd(xyz, table, fld.grid_desc)))
import ctypes as C
lm = torch.empty(16 * n * 2, device=dev)
rm = torch.empty(n * 32, device=dev)
def fwd_x(buf, mode):
    N.check(N.lib().arcn_hashgrid_fwd_xcd(xyz.data_ptr(), table.data_ptr(), C.addressof(fld.grid_desc), buf.data_ptr(), mode, n, n, None, N.stream()))
print('fwd XCD level-major %.3f ms' % timeit(lambda: fwd_x(lm, 1)))
print('fwd XCD row-major   %.3f ms' % timeit(lambda: fwd_x(rm, 0)))
ref = F.hashgrid_fwd(xyz, table, fld.grid_desc)
print('LM == row-major:', torch.equal(lm.view(16, n, 2).permute(1, 0, 2).reshape(n, 32), ref), ' RM ==', torch.equal(rm.view(n, 32), ref))
for l in range(16):
    desc = N.make_hashgrid_desc([fld.resolutions[l]], [fld.offsets[l], fld.offsets[l + 1]], 2, fld.min_xyz, fld.max_xyz)
    d1 = torch.randn(n, 2, device=dev)
    tb = timeit(lambda: F.hashgrid_bwd(xyz, table, d1, desc, dtable=dt, workspace=ws))
    tf = timeit(lambda: F.hashgrid_fwd(xyz, table, desc))
    print('level %2d res %4d: bwd %.3f ms  fwd %.3f ms' % (l, fld.resolutions[l], tb, tf))
# uncontended atomic rate: random xyz in the volume
xr = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
print('random pts bwd plain %.3f ms, fwd %.3f ms' % (timeit(lambda: F.hashgrid_bwd(xr, table, dout, fld.grid_desc, dtable=dt)),
      timeit(lambda: F.hashgrid_fwd(xr, table, fld.grid_desc))))
