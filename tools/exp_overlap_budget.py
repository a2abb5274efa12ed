#!/usr/bin/env python
"""What the second stream buys the headline step: the step with the marching of a later batch beside it (depth 2, the bench's schedule; depth 1),
the step with its own marching inline on the step's stream (no overlap at all), and the marching chain alone.   python tools/exp_overlap_budget.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays  # noqa: E402

dev = torch.device('cuda:0')
pool = [synthetic_rays(8320, seed=i, device=dev) for i in range(8)]
tgt = torch.rand(8320, 3, device=dev)
bits = torch.from_numpy(synthetic_bitfield(128, 0.05, seed=0))


def make(depth):
    cfg = NgpConfig()
    pipe = NgpPipeline(NgpField(cfg, device=dev, seed=0), max_rays=32768, max_samples=1 << 20, packed_bits=True, prefetch_depth=depth)
    pipe.set_bitfield(bits)
    return pipe


def timed(fn, n=256, warm=32):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(warm + i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for rep in range(2):
    p2, p1, p0 = make(2), make(1), make(1)
    t2 = timed(lambda i: p2.train_step(pool[i % 8][0], pool[i % 8][1], tgt, next_rays=pool[(i + 2) % 8]))
    t1 = timed(lambda i: p1.train_step(pool[i % 8][0], pool[i % 8][1], tgt, next_rays=pool[(i + 1) % 8]))
    t0 = timed(lambda i: p0.train_step(pool[i % 8][0], pool[i % 8][1], tgt))
    sp = p0._sets[1]

    def chain(i):
        p0._sample_into(sp, pool[i % 8][0], pool[i % 8][1])
        sp['noise'].normal_(0.0, 1.0)
    tc = timed(chain)
    print('two batches ahead %.4f ms   one batch ahead %.4f   marching inline (no overlap) %.4f   the marching chain alone %.4f   => the step without any marching %.4f'
          % (t2, t1, t0, tc, t0 - tc))
