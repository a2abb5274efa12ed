"""experiment: the marcher alone on the bench ray distribution"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays
dev = torch.device('cuda:0')
cfg = NgpConfig()
fld = NgpField(cfg, device=dev)
pipe = NgpPipeline(fld, max_rays=32768, max_samples=1 << 20)
pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, 0)))
o, d = synthetic_rays(8320, seed=0, device=dev)
for _ in range(3):
    pipe.sample(o, d)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    pipe.sample(o, d)
e1.record(); torch.cuda.synchronize()
c = pipe.buf['counts'][:8320]
print('sample() %.1f us; samples %d; rays with samples %d; max per ray %d' % (e0.elapsed_time(e1) / 20 * 1e3, int(pipe.n_dev.item()), int((c > 0).sum()), int(c.max())))
