"""latency floor of the fused compositor: time arcn_composite_packed_train on the bench batch for the first R rays, R = 64 .. 8320"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arcnerf_amd import _native as N
from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays
dev = torch.device('cuda:0')
cfg = NgpConfig(); fld = NgpField(cfg, device=dev)
pipe = NgpPipeline(fld, max_rays=32768, max_samples=1 << 20)
pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, 0)))
o, d = synthetic_rays(8320, seed=0, device=dev)
tgt = torch.rand(8320, 3, device=dev)
pipe.forward(o, d, None, train=True, noise='auto', huber_target=tgt)
b = pipe.buf
L, st = N.lib(), N.stream()
def run(R):
    N.check(L.arcn_composite_packed_train(N.ptr(b['sigma']), N.ptr(b['rgb_s']), N.ptr(b['t']), N.ptr(b['offsets']), N.ptr(b['noise']), None, 0, R, 2,
                                          b['p_dense'].data_ptr(), 0, 0, N.ptr(tgt), 0.1, 3000.0, N.ptr(b['rgb']), N.ptr(b['depth']), N.ptr(b['mask']),
                                          N.ptr(b['d_rgb']), b['loss_ring'][0].data_ptr(), N.ptr(b['d_sigma']), N.ptr(b['d_rgb_s']), N.ptr(b['counts']), st), 'x')
for R in (64, 512, 2048, 8320):
    run(R); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run(R)
    e1.record(); torch.cuda.synchronize()
    print('R %5d  %.1f us per launch' % (R, e0.elapsed_time(e1) / 50 * 1e3))
