"""G16: MultiVol (cascaded-volume model) vectors from the reference, run in the build container only.

The reference's MultiVol samples through its CUDA-only `_multivol_func` extension, which cannot run here.  This script runs the
REST of the reference model on CPU — ray / outer-volume bounds, trimming to the longest ray, gathering the valid samples,
the nets, padding with each ray's last sample, compositing, FullModel's output handling, autograd — with the sampler call
replaced by this repo's CPU oracle of that kernel (oracle/src/orc_bitfield.c, pcg32 state = the extension's file-static
generator at its first launch).  The sampled zvals / mask are stored too, so the fixture also serves as a vector for the HIP
sampler.  configs/models/multivol.yaml with small grids and torch-Linear nets (GeoNet / RadianceNet, frequency encoders), both
`inclusive` settings.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, '/root/reference')
sys.path.insert(1, ROOT)
_r = types.ModuleType('pytorch3d.transforms.rotation_conversions')
for _n in ['axis_angle_to_matrix', 'matrix_to_axis_angle', 'matrix_to_rotation_6d', 'rotation_6d_to_matrix']:
    setattr(_r, _n, lambda *a, **k: None)
sys.modules['pytorch3d'] = types.ModuleType('pytorch3d')
sys.modules['pytorch3d.transforms'] = types.ModuleType('pytorch3d.transforms')
sys.modules['pytorch3d.transforms.rotation_conversions'] = _r
import warnings  # noqa: E402

warnings.filterwarnings('ignore')
import numpy as np  # noqa: E402
import torch  # noqa: E402

import arcnerf.ops.multivol_func as ref_ops  # noqa: E402

ref_ops.CUDA_BACKEND_AVAILABLE = True
import arcnerf.models.multivol_bkg_model as ref_mv  # noqa: E402
from arcnerf.models import build_model  # noqa: E402
from common.utils.cfgs_utils import load_configs  # noqa: E402
from oracle import oracle as orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
N_GRID, N_CASCADE, N_SAMPLE = 16, 3, 96
OVERRIDES = ['--model.basic_volume.n_grid', str(N_GRID), '--model.basic_volume.n_cascade', str(N_CASCADE),
             '--model.rays.n_sample', str(N_SAMPLE), '--model.rays.cone_angle', '0.0625', '--model.rays.noise_std', '0.0',
             '--model.chunk_pts', '4096', '--model.optim.near_distance', '0.05',
             '--model.geometry.type', 'GeoNet', '--model.geometry.W', '32', '--model.geometry.D', '2', '--model.geometry.W_feat', '16',
             '--model.geometry.encoder.type', 'FreqEmbedder', '--model.geometry.encoder.n_freqs', '4',
             '--model.geometry.encoder.include_input', 'True', '--model.geometry.encoder.backend', 'None',
             '--model.radiance.type', 'RadianceNet', '--model.radiance.W', '32', '--model.radiance.D', '1',
             '--model.radiance.encoder.view.type', 'FreqEmbedder', '--model.radiance.encoder.view.n_freqs', '2',
             '--model.radiance.encoder.view.include_input', 'True', '--model.radiance.encoder.view.backend', 'None']

_state = {}


def oracle_sampler(rays_o, rays_d, near, far, n_pts, cone_angle, min_step, max_step, min_aabb_range, aabb_range, n_grid, n_cascade,
                   bitfield, near_distance=0.0, inclusive=False):
    h = _state['rng']
    z, m, c = orc.sparse_sampling_in_multivol_bitfield(
        rays_o.numpy(), rays_d.numpy(), near.numpy(), far.numpy(), n_pts, cone_angle, min_step, max_step,
        min_aabb_range.permute(1, 0).contiguous().numpy(), aabb_range.permute(1, 0).contiguous().numpy(), n_grid, n_cascade,
        bitfield.numpy(), near_distance, inclusive, h.state, h.inc)
    h.advance()
    _state.setdefault('samples', []).append((near.numpy().copy(), far.numpy().copy(), z.copy(), m.copy()))
    return torch.from_numpy(z), torch.from_numpy(m)


ref_mv.CUDA_BACKEND_AVAILABLE = True
ref_mv.sparse_sampling_in_multivol_bitfield = oracle_sampler


def main():
    g = torch.Generator().manual_seed(1616)
    B, N = 1, 96
    o = (torch.rand(B, N, 3, generator=g) - 0.5) * 0.6                      # cameras inside the inner volume (side 1)
    o[:, N // 2:] = (torch.rand(B, N - N // 2, 3, generator=g) - 0.5) * 3.0  # ... and some between the cascades
    d = torch.randn(B, N, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    inputs = {'rays_o': o, 'rays_d': d, 'rays_r': torch.zeros(B, N, 1), 'img': torch.rand(B, N, 3, generator=g)}
    out = {}
    rng = np.random.default_rng(16)
    for tag, inclusive in (('incl_', True), ('excl_', False)):
        torch.manual_seed(1617)
        ov = list(OVERRIDES) + ['--model.basic_volume.inclusive', str(inclusive)]
        model = build_model(load_configs('/root/reference/configs/models/multivol.yaml', ov), None)
        mv = model.fg_model
        levels = N_CASCADE if inclusive else N_CASCADE - 1
        bits = (rng.random(N_GRID ** 3 * levels) < 0.35)
        with torch.no_grad():
            mv.density_bitfield.copy_(torch.from_numpy(np.packbits(bits, bitorder='little')))
            for n_, p in model.named_parameters():
                if n_.endswith('geo_net.layers.2.weight') or n_.endswith('geo_net.layers.4.weight'):
                    p[:1] += 0.25      # some density everywhere
        for k, v in model.state_dict().items():
            out[tag + 'sd.' + k] = v.numpy()
        _state['rng'] = orc.Pcg32(9121)
        _state['samples'] = []
        with torch.no_grad():
            res = model({k: v.clone() for k, v in inputs.items()}, inference_only=True)
        for k, v in res.items():
            out[tag + 'infer_' + k] = v.numpy()
        res = model({k: v.clone() for k, v in inputs.items()}, inference_only=False)
        for k, v in res.items():
            if torch.is_tensor(v):
                out[tag + 'train_' + k] = v.detach().numpy()
        rgb_key = [k for k in res if k.startswith('rgb')][0]
        loss = ((res[rgb_key] - inputs['img']) ** 2).mean()
        loss.backward()
        out[tag + 'train_loss'] = loss.detach().numpy()
        for k, p in model.named_parameters():
            if p.grad is not None:
                out[tag + 'grad.' + k] = p.grad.numpy()
        for call, (nr, fr, z, m) in enumerate(_state['samples']):
            out[tag + 'call{}_near'.format(call)], out[tag + 'call{}_far'.format(call)] = nr, fr
            out[tag + 'call{}_zvals'.format(call)], out[tag + 'call{}_mask'.format(call)] = z, m
        print(tag, {k: tuple(v.shape) for k, v in res.items() if torch.is_tensor(v)}, 'samples/ray',
              [float(s[3].sum(1).mean()) for s in _state['samples']], 'empty rays', [int((s[3].sum(1) == 0).sum()) for s in _state['samples']],
              'loss', float(loss))
    for k, v in inputs.items():
        out['in_' + k] = v.numpy()
    out['overrides'] = np.array(OVERRIDES)
    path = os.path.join(OUT, 'g16_multivol_model.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) / 1024, 'KB')


if __name__ == '__main__':
    main()
