"""K11 cascade marcher timing on two ray sets: cameras inside the (excluded) inner volume / cameras outside it looking in"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arcnerf_amd.ops import functional as F
from arcnerf_amd.pipeline import synthetic_rays
dev = torch.device('cuda:0')
n_grid, n_cascade, n_pts = 128, 5, 1024
def make_bits(levels):
    bits = (torch.rand(levels * n_grid ** 3, device=dev) < 0.03).view(-1, 8)
    return (bits.to(torch.int32) * (2 ** torch.arange(8, device=dev)).to(torch.int32)).sum(-1).to(torch.uint8)
inner = torch.tensor([[-0.5] * 3, [0.5] * 3], device=dev)
outer = inner * 2 ** (n_cascade - 1)
min_step, max_step = 3 ** 0.5 / n_pts, 3 ** 0.5 * 16 / n_grid
import itertools
for (name, radius), incl in itertools.product((('inside', 0.3), ('outside', 2.0)), (False, True)):
    o, d = synthetic_rays(4096, seed=1, device=dev, radius=radius)
    near, far, _, _ = F.aabb_intersection_torch(o, d, torch.stack([outer[0], outer[1]], -1)[None], 1e-7)
    bf = make_bits(n_cascade if incl else n_cascade - 1)
    def run():
        return F.sparse_sampling_in_multivol_bitfield(o, d, near, far, n_pts, 1.0 / 256, min_step, max_step, inner, outer, n_grid, n_cascade,
                                                      bf, 0.05, incl, 1234567, 3, want_counts=True)
    z, m, c = run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    print('incl=%d %-8s %.3f ms per call (incl. output allocation), mean samples/ray %.1f, checksum %d' % (incl, name, e0.elapsed_time(e1) / 10, float(c.float().mean()), int(c.sum())))
