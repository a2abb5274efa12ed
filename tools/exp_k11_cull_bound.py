"""Upper bound of what ray culling could buy the cascade marcher (K11) on config 4's bench batch: the kernel on all 4096 rays against the kernel on
only the rays that produce samples"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcnerf_amd.models import build_model
from arcnerf_amd.ops import functional as Fn
from arcnerf_amd.ops.multivol_func import multivol_rng
from arcnerf_amd.pipeline import synthetic_cascade_bits, synthetic_rays
from arcnerf_amd.utils.cfgs_utils import load_configs
dev = torch.device('cuda:0')
m = build_model(load_configs('configs/neus_ngp_multivol.yaml', [])).to(dev)
b = m.bkg_model
b.density_bitfield.copy_(torch.from_numpy(synthetic_cascade_bits(128, b.n_levels, 0.05, seed=5)).to(dev))
o, d = synthetic_rays(4096, seed=0, device=dev)
near, far = b.get_near_far_from_rays(o, d)
rng = multivol_rng()


def run(o, d, near, far):
    return Fn.sparse_sampling_in_multivol_bitfield(o, d, near, far, b.get_ray_cfgs('n_sample'), b.cone_angle, b.min_step, b.max_step, b.basic_volume.get_range23(),
                                                   b.max_volume.get_range23(), b.n_grid, b.n_cascade, b.density_bitfield, b.get_optim_cfgs('near_distance'), b.inclusive,
                                                   rng.state, rng.inc, want_counts=True, dense=False)


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


_, _, counts = run(o, d, near, far)
hit = counts > 0
print('rays with samples: %d of %d, samples %d' % (int(hit.sum()), hit.numel(), int(counts.sum())))
print('all rays      : %.1f us' % t(lambda: run(o, d, near, far)))
oh, dh, nh, fh = o[hit].contiguous(), d[hit].contiguous(), near[hit].contiguous(), far[hit].contiguous()
print('hit rays only : %.1f us' % t(lambda: run(oh, dh, nh, fh)))
om, dm, nm, fm = o[~hit].contiguous(), d[~hit].contiguous(), near[~hit].contiguous(), far[~hit].contiguous()
print('miss rays only: %.1f us' % t(lambda: run(om, dm, nm, fm)))
