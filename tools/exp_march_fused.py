"""the marcher chain alone on the bench batch: three-pass form (march_count + scan + march_write) against arcn_march_packed"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays
dev = torch.device('cuda:0')
cfg = NgpConfig()
fld = NgpField(cfg, device=dev, seed=0)
o, d = synthetic_rays(8320, seed=1000, device=dev)
for fused in (False, True, False, True):
    pipe = NgpPipeline(fld, max_rays=32768, max_samples=1 << 20)
    pipe.march_fused = fused
    pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(cfg.n_grid, 0.05, seed=0)))
    for _ in range(3):
        pipe.sample(o, d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        pipe.sample(o, d)
    e1.record(); torch.cuda.synchronize()
    print('fused' if fused else '3-pass', 'sample chain: %.1f us' % (e0.elapsed_time(e1) / 20 * 1e3), 'samples', int(pipe.n_dev.item()))
