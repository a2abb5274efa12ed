import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
from arcnerf_amd.ops import functional as F
from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays
dev = torch.device('cuda:0')
cfg = NgpConfig(); fld = NgpField(cfg, device=dev)
pipe = NgpPipeline(fld, max_rays=32768, max_samples=1 << 20)
pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, 0)))
o, d = synthetic_rays(8320, seed=0, device=dev)
pipe.forward(o, d, None, train=True)
n = int(pipe.n_dev.item()); xyz = pipe.buf['xyz'][:n].contiguous(); table = fld.view('table').view(-1, 2)
gdx = torch.randn(n, 3, device=dev); dout = torch.randn(n, 32, device=dev)
def timeit(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
ws = F.hashgrid_bwd_workspace(fld.grid_desc, 2 * n, dev)
print('samples', n)
print('second-order table scatter, atomics %.3f ms' % timeit(lambda: F.hashgrid_bwd_bwd(xyz, gdx, table, dout, fld.grid_desc, want_ddout=False, workspace=None)))
print('second-order table scatter, binned  %.3f ms' % timeit(lambda: F.hashgrid_bwd_bwd(xyz, gdx, table, dout, fld.grid_desc, want_ddout=False, workspace=ws)))
print('second-order gather (ddout + d2x)   %.3f ms' % timeit(lambda: F.hashgrid_bwd_bwd(xyz, gdx, table, dout, fld.grid_desc, want_dtable=False, want_d2xyz=True)))
a = F.hashgrid_bwd_bwd(xyz, gdx, table, dout, fld.grid_desc, want_ddout=False, workspace=None)[1]
b = F.hashgrid_bwd_bwd(xyz, gdx, table, dout, fld.grid_desc, want_ddout=False, workspace=ws)[1]
print('max rel diff', float((a - b).abs().max() / a.abs().max()))
