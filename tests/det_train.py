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
cfg = NgpConfig()
fld = NgpField(cfg, device=dev, seed=0)
R, NB = 4096, 8
pipe = NgpPipeline(fld, max_rays=R, max_samples=1 << 20, prefetch_depth=2)
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
for i in range(steps):
    o, d, tgt, bkg = pool[i % NB]
    nxt = pool[(i + pipe.prefetch_depth) % NB]
    loss = pipe.train_step(o, d, tgt, bkg_color=bkg, next_rays=(nxt[0], nxt[1]))
    pipe.update_occupancy(i + 1, apply=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
gmax, overflowed = F.hashgrid_bwd_status(fld.grid_desc, pipe.cap, pipe.hash_ws)
print(json.dumps({'sha256': hashlib.sha256(fld.params.cpu().numpy().tobytes()).hexdigest(), 'loss': float(loss), 'steps': steps,
                  'deterministic': F.deterministic(), 'scatter_overflowed': overflowed, 'scatter_gmax': gmax,
                  'ms_per_step': dt / steps * 1e3, 'occupied': float(pipe.bitfield.float().mean())}))
