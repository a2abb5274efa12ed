#!/usr/bin/env python
"""What state of the memory system costs the hash-grid gather its 14 us inside the step?  The gather on the bench's batch, timed (events)
after: nothing (back to back), a rewrite of the table by another kernel (the plain Adam pass: the table and its moments read and written,
like the scatter's consumer does at the end of a step), 1 GB streamed through the caches (a device copy), both, and a READ of the table
after the rewrite (a sum over it) before the gather.   python tools/exp_gather_state.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from arcnerf_amd import _native as N  # noqa: E402
from arcnerf_amd.ops import functional as F  # noqa: E402
from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays  # noqa: E402

dev = torch.device('cuda:0')
cfg = NgpConfig()
fld = NgpField(cfg, device=dev, seed=0)
pipe = NgpPipeline(fld, max_rays=32768, max_samples=1 << 20, packed_bits=True, prefetch_depth=2)
pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(cfg.n_grid, 0.05, seed=0)))
o, d = synthetic_rays(8320, seed=1000, device=dev)
pipe.sample(o, d)
b, S = pipe.buf, pipe.cap
L, st = N.lib(), N.stream()
table = fld.view('table')
big_a = torch.empty(1 << 27, dtype=torch.float32, device=dev)   # 512 MB
big_b = torch.empty(1 << 27, dtype=torch.float32, device=dev)
step = [0]


def gather():
    N.check(L.arcn_hashgrid_fwd_xcd(N.ptr(b['xyz']), N.ptr(table), N.C.addressof(fld.grid_desc), N.ptr(b['feat']), 1, S, S, pipe.n_dev.data_ptr(), st), 'fwd')


def rewrite():
    step[0] += 1
    F.adam_ema_step(fld.params, fld.grads, pipe.exp_avg, pipe.exp_avg_sq, fld.params, step[0], lr=1e-6, ema_decay=0.95, zero_grad=True)


def stream():
    big_b.copy_(big_a)


def read_table():
    return table.sum()


tgt = torch.rand(8320, 3, device=dev)


def train_step():
    pipe.train_step(o, d, tgt)


pool = [synthetic_rays(8320, seed=2000 + i, device=dev) for i in range(64)]
turn = [0]


def train_step_marching():
    """... with the marching of another batch issued behind the step's tail on the second stream (the bench's schedule): it runs beside the gather"""
    turn[0] += 1
    pipe.train_step(o, d, tgt, next_rays=pool[turn[0] % len(pool)])


def timed(pre, iters=40):
    ts = []
    for _ in range(iters + 5):
        for f in pre:
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gather()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[5:])
    return ts[len(ts) // 2], ts[0]


for name, pre in (('back to back', []), ('after a rewrite of the table (Adam pass)', [rewrite]), ('after 1 GB streamed (copy)', [stream]),
                  ('after stream, then rewrite', [stream, rewrite]), ('after rewrite, then stream', [rewrite, stream]),
                  ('after rewrite, then a read of the table (sum)', [rewrite, read_table]), ('after a whole training step (no marcher beside it)', [train_step]), ('after a training step, then a read of the table', [train_step, read_table]),
                  ('after a training step WITH the next marching issued behind it', [train_step_marching]),
                  ('back to back again', [])):
    med, mn = timed(pre)
    print('%-52s median %6.1f us   min %6.1f us' % (name, med, mn))
