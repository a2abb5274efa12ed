#!/usr/bin/env python
"""Which kernel of the marching chain costs the hash-grid gather its 18 us when they run side by side?  The gather on the main stream, ONE
kind of kernel of the chain on the second stream beside it (repeated to cover the gather's duration), events around the gather.
    python tools/exp_gather_corun.py"""
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
pipe = NgpPipeline(fld, max_rays=32768, max_samples=1 << 20, packed_bits=True, prefetch_depth=2)
pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(cfg.n_grid, 0.05, seed=0)))
o, d = synthetic_rays(8320, seed=1000, device=dev)
o2, d2 = synthetic_rays(8320, seed=1001, device=dev)
pipe.sample(o, d)
b, S, R = pipe.buf, pipe.cap, 8320
sp = pipe._sets[2]          # a spare set for the side stream's kernels
L = N.lib()
table = fld.view('table')
aux = torch.cuda.Stream(device=dev)
pipe._sample_into(sp, o2, d2)   # fills the spare set once (counts, offsets, t, ray_id)
torch.cuda.synchronize()


def gather():
    N.check(L.arcn_hashgrid_fwd_xcd(N.ptr(b['xyz']), N.ptr(table), N.C.addressof(fld.grid_desc), N.ptr(b['feat']), 1, S, S, pipe.n_dev.data_ptr(), N.stream()), 'fwd')


def k_march():
    N.check(L.arcn_march_count_culled(N.ptr(o2), N.ptr(d2), N.ptr(pipe.aabb23), cfg.n_grid, N.ptr(pipe._occ()), int(pipe.packed_bits), N.ptr(pipe._coarse),
                                      cfg.n_sample, cfg.dt, cfg.near_distance, int(pipe.torch_aabb), pipe.rng.state, pipe.rng.inc, N.ptr(sp['scratch_t']),
                                      N.ptr(sp['counts']), N.ptr(sp['near']), N.ptr(sp['far']), R, N.stream()), 'march')


def k_scan():
    N.check(L.arcn_exclusive_scan_i32(N.ptr(sp['counts']), N.ptr(sp['offsets']), R, S, N.ptr(sp['p_dense']), N.stream()), 'scan')


def k_write():
    N.check(L.arcn_march_write(N.ptr(sp['scratch_t']), N.ptr(sp['counts']), N.ptr(sp['offsets']), cfg.n_sample, N.ptr(sp['t']), N.ptr(sp['ray_id']), R, S, N.stream()), 'write')


def k_points():
    N.check(L.arcn_packed_points(N.ptr(o2), N.ptr(d2), N.ptr(sp['t']), N.ptr(sp['ray_id']), N.ptr(sp['xyz']), N.ptr(sp['dirs']), S, sp['offsets'][R:R + 1].data_ptr(), N.stream()), 'points')


def k_sh():
    N.check(L.arcn_ngp_ray_sh(N.ptr(d2), cfg.sh_degree, N.ptr(sp['sh_ray']), R, N.stream()), 'sh')


def k_noise():
    sp['noise'].normal_(0.0, 1.0)


def k_chain():
    k_march(); k_scan(); k_write(); k_points(); k_sh(); k_noise()


def alone(fn, reps, iters=30):
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(ts)[len(ts) // 2]


def beside(fn, reps, iters=40):
    ts = []
    main = torch.cuda.current_stream()
    for _ in range(iters + 5):
        torch.cuda.synchronize()
        if fn is not None:
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                for _ in range(reps):
                    fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gather()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[5:])
    return ts[len(ts) // 2]


print('gather alone: %.1f us' % beside(None, 0))
for name, fn in (('march_count_culled', k_march), ('exclusive_scan', k_scan), ('march_write', k_write), ('packed_points', k_points), ('ngp_ray_sh', k_sh),
                 ('noise normal_ (2^20 floats)', k_noise), ('the whole chain', k_chain)):
    t = alone(fn, 4)
    reps = max(1, int(round(70.0 / max(t, 1.0))))
    print('%-30s alone %6.1f us per launch; gather beside %2d of them: %6.1f us' % (name, t, reps, beside(fn, reps)))
print('gather alone again: %.1f us' % beside(None, 0))
