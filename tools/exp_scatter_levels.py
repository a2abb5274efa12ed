#!/usr/bin/env python
"""The binned scatter of the bench's batch one LEVEL at a time (arcn_hashgrid_bwd_lm_levels with a one-bit mask; the workgroups of the other
levels leave at once), producer and consumer timed separately by events around ... the whole call; run under rocprofv3 --kernel-trace
--stats for the split.  Without rocprof: time of the pair per level.     python tools/exp_scatter_levels.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from arcnerf_amd import _native as N  # noqa: E402
from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays  # noqa: E402

dev = torch.device('cuda:0')
cfg = NgpConfig()
fld = NgpField(cfg, device=dev, seed=0)
pipe = NgpPipeline(fld, max_rays=32768, max_samples=1 << 20, packed_bits=True)
pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(cfg.n_grid, 0.05, seed=0)))
o, d = synthetic_rays(8320, seed=0, device=dev)
tgt = torch.rand(8320, 3, device=dev)
pipe.train_step(o, d, tgt)          # fills xyz / d_feat of the batch
torch.cuda.synchronize()
b, S = pipe.buf, pipe.cap
L, st = N.lib(), N.stream()
n = int(pipe.n_dev.item())


def run(mask):
    N.check(L.arcn_hashgrid_bwd_lm_levels(N.ptr(b['xyz']), N.ptr(b['d_feat']), S, N.C.addressof(fld.grid_desc), N.ptr(fld.view('table', fld.grads)),
                                          N.ptr(pipe.hash_ws), pipe.hash_ws.numel(), S, pipe.n_dev.data_ptr(), int(mask), 0, st), 'bwd_levels')


def timed(mask, iters=20):
    for _ in range(3):
        run(mask)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run(mask)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print('samples', n, ' resolutions', fld.resolutions)
print('all levels: %.1f us' % timed(0xffff))
tot = 0.0
for l in range(0 if os.environ.get('GROUPS_ONLY') else cfg.n_levels):
    t = timed(1 << l)
    tot += t
    print('level %2d (res %4d, %7d rows): %6.1f us' % (l, fld.resolutions[l], fld.offsets[l + 1] - fld.offsets[l], t))
print('sum of the single-level calls: %.1f us (each pays the launch pair and the counter fill again)' % tot)
for name, mask in (('dense 0-4', 0x1f), ('hashed 5-9', 0x3e0), ('hashed 10-15', 0xfc00)):
    print('%-14s %6.1f us' % (name, timed(mask)))
