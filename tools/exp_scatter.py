"""experiment: time only the binned hash-grid scatter on the bench sample distribution (run under rocprofv3 for the split)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arcnerf_amd.ops import functional as F
from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays

dev = torch.device('cuda:0')
cfg = NgpConfig()
fld = NgpField(cfg, device=dev)
pipe = NgpPipeline(fld, max_rays=32768, max_samples=1 << 20)
pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, 0)))
o, d = synthetic_rays(8320, seed=0, device=dev)
pipe.forward(o, d, None, train=True)
n = int(pipe.n_dev.item())
xyz = pipe.buf['xyz'][:n].contiguous()
table = fld.view('table')
dt = torch.zeros_like(table)
dout = torch.randn(n, 32, device=dev)
ws = F.hashgrid_bwd_workspace(fld.grid_desc, n, dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    F.hashgrid_bwd(xyz, table, dout, fld.grid_desc, dtable=dt, workspace=ws)
e0.record()
for _ in range(20):
    F.hashgrid_bwd(xyz, table, dout, fld.grid_desc, dtable=dt, workspace=ws)
e1.record(); torch.cuda.synchronize()
print('samples %d  scatter %.3f ms' % (n, e0.elapsed_time(e1) / 20))
