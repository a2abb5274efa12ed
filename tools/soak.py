"""soak: a few thousand steps of the full NGP config on a synthetic scene with the occupancy refresh APPLIED (async stream),
rotating ray batches with prefetch; checks finiteness, loss trend, bitfield evolution and buffer bounds"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays
dev = torch.device('cuda:0')
cfg = NgpConfig()
fld = NgpField(cfg, device=dev, seed=0)
pipe = NgpPipeline(fld, max_rays=8320, max_samples=1 << 20)
truth = torch.from_numpy(synthetic_bitfield(cfg.n_grid, 0.05, 0)).to(dev)
R, NB = 8320, 8
pool = []
probe = NgpPipeline(NgpField(cfg, device=dev, seed=1), max_rays=R, max_samples=1 << 20)
probe.set_bitfield(truth)
for k in range(NB):
    o, d = synthetic_rays(R, seed=100 + k, device=dev)
    probe.sample(o, d)
    hit = (probe.buf['counts'][:R] > 1).float()[:, None]
    tgt = (hit * torch.tensor([0.8, 0.3, 0.1], device=dev) + (1 - hit) * 1.0).contiguous()   # white background
    pool.append((o, d, tgt, torch.ones(R, 3, device=dev)))
del probe
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
losses, fracs, maxs = [], [], []
t0 = time.perf_counter()
for i in range(steps):
    o, d, tgt, bkg = pool[i % NB]
    nxt = pool[(i + 1) % NB]
    loss = pipe.train_step(o, d, tgt, bkg_color=bkg, next_rays=(nxt[0], nxt[1]))
    pipe.update_occupancy(i + 1, apply=True)
    if i % 250 == 0 or i == steps - 1:
        losses.append(float(loss)); fracs.append(float(pipe.bitfield.float().mean())); maxs.append(int(pipe.n_dev.item()))
        print('step %5d loss %.4f occupied %.4f samples %d' % (i, losses[-1], fracs[-1], maxs[-1]), flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
assert all(l == l and l < 1e9 for l in losses), losses
assert torch.isfinite(fld.params).all() and torch.isfinite(pipe.ema).all()
assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])
print('OK: %d steps in %.1f s (%.3f ms/step incl. host prints), loss %.4f -> %.4f, occupied %.3f -> %.3f' %
      (steps, dt, dt / steps * 1e3, losses[0], losses[-1], fracs[0], fracs[-1]))
