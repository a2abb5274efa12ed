"""Run by tests/test_gpu_deterministic.py (and tools) in a subprocess: the full NGP configuration trained for N steps on a synthetic scene
as tools/soak.py does - prefetch, density noise, occupancy refresh APPLIED on its own stream - then prints a sha256 of the parameter
buffer, the final loss and the scatter's overflow flag as one JSON line."""
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from arcnerf_amd.ops import functional as F  # noqa: E402
from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device('cuda:0')
white = os.environ.get('DET_WHITE') == '1'
cfg = NgpConfig(white_bkg=white)
if os.environ.get('DET_POLLUTE') == '1':      # fill the caching allocator with garbage first: an uninitialised read would then differ from run to run
    junk = [torch.randn(64 << 20, device=dev) * float(time.time() % 7 + 1) for _ in range(6)]
    torch.cuda.synchronize()
    del junk
fld = NgpField(cfg, device=dev, seed=0)
R, NB = (32768 if os.environ.get('DET_BIG') == '1' else 4096), 8
pipe = NgpPipeline(fld, max_rays=R, max_samples=1 << 20, prefetch_depth=1 if os.environ.get('DET_DEPTH1') == '1' else 2)
truth = torch.from_numpy(synthetic_bitfield(cfg.n_grid, 0.05, 0)).to(dev)
probe = NgpPipeline(NgpField(cfg, device=dev, seed=1), max_rays=R, max_samples=1 << 20)
probe.set_bitfield(truth)
pool = []
for k in range(NB):
    o, d = synthetic_rays(R, seed=100 + k, device=dev)
    probe.sample(o, d)
    hit = (probe.buf['counts'][:R] > 1).float()[:, None]
    pool.append((o, d, (hit * torch.tensor([0.8, 0.3, 0.1], device=dev) + (1 - hit) * 1.0).contiguous(), torch.ones(R, 3, device=dev)))
del probe
torch.manual_seed(0)        # the density noise of the training steps
torch.cuda.synchronize()
t0 = time.perf_counter()
loss = None
dyn = os.environ.get('DET_DYN') == '1'            # dynamic batch size like tools/psnr_curve.py (one host read every 16 steps)
no_prefetch = os.environ.get('DET_NOPREFETCH') == '1'
no_occ = os.environ.get('DET_NOOCC') == '1'
n_rays = 512 if dyn else R
for i in range(steps):
    o, d, tgt, bkg = pool[i % NB]
    nxt = pool[(i + pipe.prefetch_depth) % NB]
    loss = pipe.train_step(o[:n_rays], d[:n_rays], tgt[:n_rays], bkg_color=None if white else bkg[:n_rays],
                           next_rays=None if no_prefetch else (nxt[0][:n_rays], nxt[1][:n_rays]))
    if not no_occ:
        pipe.update_occupancy(i + 1, apply=True)
    if dyn and (i + 1) % 16 == 0:
        sm = max(1, pipe.sample_count())
        n_rays = int(min(R, max(128, (int(n_rays * (1 << 18) / sm) + 127) // 128 * 128)))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
gmax, overflowed = F.hashgrid_bwd_status(fld.grid_desc, pipe.cap, pipe.hash_ws)
print(json.dumps({'sha256': hashlib.sha256(fld.params.cpu().numpy().tobytes()).hexdigest(), 'loss': float(loss), 'steps': steps,
                  'deterministic': F.deterministic(), 'scatter_overflowed': overflowed, 'scatter_gmax': gmax,
                  'ms_per_step': dt / steps * 1e3, 'occupied': float(pipe.bitfield.float().mean())}))
