"""arcn_ngp_step_tail alone, piece by piece (GPU box): python tools/exp_step_tail.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from arcnerf_amd import _native as N
from arcnerf_amd.ops import functional as F
from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline

dev = torch.device('cuda:0')
cfg = NgpConfig()
fld = NgpField(cfg, device=dev, seed=0)
pipe = NgpPipeline(fld, max_rays=32768, max_samples=1 << 20, packed_bits=True, prefetch_depth=2)
S, L, t, b = pipe.cap, N.lib(), pipe._tail, pipe.buf
print('runs', t['runs'], 'clear_words', t['clear_words'], 'S', S)


def tail(runs, clear_words):
    flat = (C.c_int64 * 8)(*([v for a, b_ in runs for v in (int(a), int(b_) - int(a))] + [0] * (8 - 2 * len(runs))))
    N.check(L.arcn_ngp_step_tail(C.addressof(fld.geo_desc), N.ptr(b['geo_scratch']), t['geo_w'], C.addressof(fld.rad_desc), N.ptr(b['rad_scratch']),
                                 t['rad_w'], S, S, N.ptr(fld.params), N.ptr(fld.grads), N.ptr(pipe.exp_avg), N.ptr(pipe.exp_avg_sq), N.ptr(pipe.ema),
                                 C.cast(flat, C.c_void_p), len(runs), 1e-2, 0.9, 0.99, 1e-15, 0.0, 0.95, 1.0, 1, 1, N.ptr(pipe.hash_ws), clear_words,
                                 N.stream()), 'tail')


def separate():
    N.check(L.arcn_mlp_bwd_reduce(C.addressof(fld.geo_desc), N.ptr(b['geo_scratch']), N.ptr(fld.grads[t['geo_w']:]), S, S, N.stream()), 'r')
    N.check(L.arcn_mlp_bwd_reduce(C.addressof(fld.rad_desc), N.ptr(b['rad_scratch']), N.ptr(fld.grads[t['rad_w']:]), S, S, N.stream()), 'r')
    F.adam_ema_step_runs(fld.params, fld.grads, pipe.exp_avg, pipe.exp_avg_sq, pipe.ema, pipe._adam_rest, 1, lr=1e-2, betas=(0.9, 0.99), eps=1e-15,
                         weight_decay=0.0, ema_decay=0.95, grad_scale=1.0, zero_grad=True)
    pipe.hash_ws[:t['clear_words']].zero_()


def timeit(name, fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print('%-40s %.2f us' % (name, e0.elapsed_time(e1) / n * 1e3))


timeit('tail (all)', lambda: tail(t['runs'], t['clear_words']))
timeit('tail, no runs', lambda: tail([], t['clear_words']))
timeit('tail, no clear', lambda: tail(t['runs'], 0))
timeit('tail, reductions only', lambda: tail([], 0))
timeit('four separate launches', separate)
