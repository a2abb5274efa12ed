"""how sparse is the table gradient of one bench step?  fraction of rows of every level that received a contribution (what a sparsified
gradient all-reduce could skip) and the bytes a touched-row list would take on the wire"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays

dev = torch.device('cuda:0')
cfg = NgpConfig()
fld = NgpField(cfg, device=dev, seed=0)
pipe = NgpPipeline(fld, max_rays=32768, max_samples=1 << 20)
pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(cfg.n_grid, 0.05, seed=0)))
o, d = synthetic_rays(8320, seed=1000, device=dev)
tgt = torch.rand(8320, 3, device=dev)
rgb, _, _ = pipe.forward(o, d, None, train=True)
_, d_rgb = pipe.huber_grad(rgb, tgt)
fld.grads.zero_()
pipe.backward(o, d, d_rgb)
torch.cuda.synchronize()
g = fld.view('table', fld.grads).view(-1, 2)
tot_rows = tot_touched = 0
for l in range(cfg.n_levels):
    rows = g[fld.offsets[l]:fld.offsets[l + 1]]
    t = int((rows.abs().sum(1) > 0).sum())
    tot_rows += rows.shape[0]
    tot_touched += t
    print('level %2d rows %7d touched %7d (%.1f %%)' % (l, rows.shape[0], t, 100.0 * t / rows.shape[0]))
print('samples %d: touched %d of %d rows (%.1f %%); dense fp32 gradient %.1f MB, touched-row list (4 B index + 8 B values) %.1f MB' % (
    int(pipe.n_dev.item()), tot_touched, tot_rows, 100.0 * tot_touched / tot_rows, tot_rows * 8 / 1e6, tot_touched * 12 / 1e6))
