"""experiment: host-side cost of one train_step (submission time without waiting for the GPU) vs GPU time"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays

dev = torch.device('cuda:0')
cfg = NgpConfig()
fld = NgpField(cfg, device=dev)
pipe = NgpPipeline(fld, max_rays=8320, max_samples=1 << 19)
pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, 0)))
o, d = synthetic_rays(8320, seed=0, device=dev)
tgt = torch.rand(8320, 3, device=dev)
for _ in range(10):
    pipe.train_step(o, d, tgt, next_rays=(o, d))
torch.cuda.synchronize()
N = 50
t0 = time.perf_counter()
for _ in range(N):
    pipe.train_step(o, d, tgt, next_rays=(o, d))
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host submit %.3f ms/step, total %.3f ms/step' % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
if len(sys.argv) > 1:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(N):
        pipe.train_step(o, d, tgt, next_rays=(o, d))
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
