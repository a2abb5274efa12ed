"""Probe (round 6): does the chip have room for TWO half-batch NGP steps side by side?  One pipeline on the full batch (8320 rays) against two
independent pipelines (own parameters, own sampling streams) on 4160 rays each, issued alternately on two streams.  Not a training - each half
has its own optimiser - only the throughput question behind 'pipeline two halves of a batch through gather / nets / scatter on two streams'."""
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays  # noqa: E402

dev = torch.device('cuda:0')


def make(n_rays, seed):
    cfg = NgpConfig()
    field = NgpField(cfg, device=dev, seed=0)
    pipe = NgpPipeline(field, max_rays=32768, max_samples=1 << 20, packed_bits=True, prefetch_depth=2)
    pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(cfg.n_grid, 0.05, seed=0)))
    g = torch.Generator(device='cpu').manual_seed(seed)
    pool = []
    for i in range(8):
        o, d = synthetic_rays(n_rays, seed=seed * 100 + i, device=dev)
        pool.append((o, d, torch.rand(n_rays, 3, generator=g).to(dev), torch.rand(n_rays, 3, generator=g).to(dev)))
    return pipe, pool


def step(pipe, pool, i):
    o, d, tgt, bkg = pool[i % 8]
    nxt = pool[(i + 2) % 8]
    pipe.train_step(o, d, tgt, bkg_color=bkg, next_rays=(nxt[0], nxt[1]))


def timed(fn, steps, warm=48):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(warm + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


full, pool_f = make(8320, 1)
ms_full = timed(lambda i: step(full, pool_f, i), 200)
n_full = int(full.n_dev.item())
del full, pool_f
torch.cuda.empty_cache()
pa, pool_a = make(4160, 2)
pb, pool_b = make(4160, 3)
ms_half = timed(lambda i: step(pa, pool_a, i), 200)
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def both(i):
    with torch.cuda.stream(sa):
        step(pa, pool_a, i)
    with torch.cuda.stream(sb):
        step(pb, pool_b, i)


ms_both = timed(both, 200)
n_a, n_b = int(pa.n_dev.item()), int(pb.n_dev.item())
print('full batch   : %.4f ms/step, %d samples -> %.3e samples/s' % (ms_full, n_full, n_full / ms_full * 1e3))
print('one half     : %.4f ms/step, %d samples -> %.3e samples/s' % (ms_half, n_a, n_a / ms_half * 1e3))
print('two halves   : %.4f ms per pair, %d samples -> %.3e samples/s' % (ms_both, n_a + n_b, (n_a + n_b) / ms_both * 1e3))
