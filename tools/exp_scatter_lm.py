"""the binned scatter with LEVEL-MAJOR gradients (the training path's entry point arcn_hashgrid_bwd_lm) on the bench batch; run under
rocprofv3 --kernel-trace --stats for the producer / consumer split, with ARCN_SCATTER_LEVELS=<mask> for single levels"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arcnerf_amd import _native as N
from arcnerf_amd.ops import functional as F
from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays

dev = torch.device('cuda:0')
cfg = NgpConfig()
fld = NgpField(cfg, device=dev)
pipe = NgpPipeline(fld, max_rays=32768, max_samples=1 << 20)
pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, 0)))
o, d = synthetic_rays(8320, seed=1000, device=dev)
pipe.sample(o, d)
n = int(pipe.n_dev.item())
S = pipe.cap
b = pipe.buf
dt = torch.zeros_like(fld.view('table'))
b['d_feat'].normal_()
ws = pipe.hash_ws
L, st = N.lib(), N.stream()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def run():
    N.check(L.arcn_hashgrid_bwd_lm(N.ptr(b['xyz']), N.ptr(b['d_feat']), S, N.C.addressof(fld.grid_desc), N.ptr(dt), N.ptr(ws), ws.numel(), S,
                                   pipe.n_dev.data_ptr(), st), 'bwd_lm')


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record(); torch.cuda.synchronize()
print('samples %d  levels %s  scatter_lm %.1f us' % (n, os.environ.get('ARCN_SCATTER_LEVELS', 'all'), e0.elapsed_time(e1) / iters * 1e3))
