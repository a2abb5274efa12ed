"""the occupancy refresh of the packed NGP pipeline alone (nothing else on the GPU): ms per refresh and the gather's share.
python tools/exp_refresh.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield

dev = torch.device('cuda:0')
cfg = NgpConfig()
pipe = NgpPipeline(NgpField(cfg, device=dev, seed=0), max_rays=8320, max_samples=1 << 19)
pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(cfg.n_grid, 0.05, seed=0)))
for _ in range(3):
    pipe._refresh_occupancy(512, False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    pipe._refresh_occupancy(512, False)
e1.record()
torch.cuda.synchronize()
print('refresh alone: %.3f ms' % (e0.elapsed_time(e1) / 10))
