"""G21: the instant-ngp stack (BASELINE config 2) END TO END from the reference, run in the build container only.

`build_model(configs/models/nerf_ngp.yaml)` of the reference at the config's own dimensions (hash grid L16 F2 T2^19, base 16,
max 2048; volume 128^3; 1024 samples per ray; dt = 2*sqrt(3)/1024; near_distance 0.2), fp32, torch back-ends:
`type: FusedMLPGeoNet -> GeoNet`, `FusedMLPRadianceNet -> RadianceNet`, encoder `backend: tcnn -> torch` (tiny-cuda-nn is not
vendored).  The reference's two CUDA-only calls on this path are replaced by this repo's CPU oracle of those kernels
(oracle/src/orc_volume.c: K2 `aabb_intersection`, K3 `sparse_volume_sampling`, pcg32 state = the extension's file-static generator
at its first launch, advanced 2^32 per launch) -- exactly what make_golden_multivol.py does for K11.  Everything else is the
reference's own code: FullModel.forward -> FgModel.forward (ray culling, reduce_empty_mask, invalid-ray defaults) ->
NeRF._forward -> get_sigma_radiance_by_mask_pts (compaction, chunked nets, last-valid fill) -> HashGridEmbedder (torch) ->
GeoNet -> TruncExp -> RadianceNet(SH) -> ray_marching, and autograd for the gradients.

Two ray-bound variants: `k2` (what a reference with its CUDA extension does: K2 semantics, no eps, tmin>0) and `tb` (the
reference's own torch AABB code, near+eps / far-eps, then K3).
Two net variants: `lin` (biases, W_feat 16 -> geometry output 17 = [sigma | feat]) with add_inf_z False, and `nb` (use_bias False,
W_feat 15 -> 16 outputs; add_inf_z True).
Per variant: inference, train with noise_std 0 (outputs + gradients), train with noise_std 1 (the reference's torch.randn draw is
recorded and stored so that the same noise can be fed to the kernels).

The 48.8 MB table is NOT stored: it is `table_from_seed()` below (numpy PCG64, also used by the test), pinned by checksums.
The table gradient (12.2 M floats) is stored as: every row of levels 0..2, the rows r with r % 16 == 5 of all levels, per-level
column sums / abs sums and 4 seeded +-1 projections.
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

import arcnerf.geometry.volume as ref_volume  # noqa: E402
import arcnerf.models.base_modules.obj_bound.volume_bound as ref_vb  # noqa: E402
import arcnerf.render.ray_helper as ref_rh  # noqa: E402
from arcnerf.models import build_model  # noqa: E402
from common.utils.cfgs_utils import load_configs  # noqa: E402
from oracle import oracle as orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
N_GRID, N_SAMPLE = 128, 1024
TABLE_SEED, TABLE_AMP = 2121, 0.3
BASE = ['--model.geometry.type', 'GeoNet', '--model.geometry.encoder.backend', 'torch', '--model.geometry.encoder.dtype', 'torch.float32',
        '--model.radiance.type', 'RadianceNet', '--model.radiance.encoder.view.backend', 'torch', '--model.chunk_pts', '4096']
NETS = {
    'lin': ['--model.rays.add_inf_z', 'False'],
    'nb': ['--model.geometry.use_bias', 'False', '--model.geometry.W_feat', '15', '--model.radiance.use_bias', 'False',
           '--model.radiance.W_feat_in', '15', '--model.rays.add_inf_z', 'True'],
}
_state = {}
_orig_aabb = ref_volume.aabb_ray_intersection


def table_from_seed(n_rows, n_feat, seed=TABLE_SEED, amp=TABLE_AMP):
    """the hash table of the fixture: U(-amp, amp) from numpy's PCG64 (identical on every box; checksums in the fixture)"""
    rng = np.random.default_rng(seed)
    return ((rng.random((n_rows, n_feat), dtype=np.float32) - np.float32(0.5)) * np.float32(2.0 * amp)).astype(np.float32)


def oracle_k3(rays_o, rays_d, near, far, n_pts, dt, aabb_range, n_grid, bitfield, near_distance=0.0):
    h = _state['rng']
    aabb23 = aabb_range.permute(1, 0).contiguous().numpy()
    z, m, c = orc.sparse_volume_sampling(rays_o.numpy(), rays_d.numpy(), near.numpy(), far.numpy(), n_pts, np.float32(dt), aabb23, n_grid,
                                         bitfield.numpy(), near_distance, h.state, h.inc)
    h.advance()
    _state['samples'].append((near.numpy().copy(), far.numpy().copy(), z.copy(), m.copy()))
    return torch.from_numpy(z), torch.from_numpy(m)


def oracle_k2(rays_o, rays_d, aabb_range, eps=1e-7, force_torch=False):
    """what `aabb_ray_intersection` does on a CUDA tensor with the extension built (geometry/ray.py:291-292, ops/volume_func.py:53-66)"""
    if force_torch:
        return _orig_aabb(rays_o, rays_d, aabb_range, eps, True)
    near, far, pts, mask = orc.aabb_intersection(rays_o.numpy(), rays_d.numpy(), aabb_range.permute(0, 2, 1).contiguous().numpy())
    return torch.from_numpy(near), torch.from_numpy(far), torch.from_numpy(pts), torch.from_numpy(mask)


ref_vb.CUDA_BACKEND_AVAILABLE = True
ref_vb.sparse_volume_sampling = oracle_k3


def make_inputs():
    from arcnerf_amd.pipeline import synthetic_bitfield, synthetic_rays  # pure torch / numpy generators (data only)
    N = 144
    o, d = synthetic_rays(N, seed=2100, device='cpu', hw=90)       # a low-res camera: rays fan out over the whole volume
    g = torch.Generator().manual_seed(2101)
    d[100:124] = -d[100:124]                                       # 24 rays looking away: miss the box
    o[124:134] = (torch.rand(10, 3, generator=g) - 0.5) * 1.2      # 10 cameras INSIDE the volume (near = 0 -> near_distance)
    tilt = torch.randn(10, 3, generator=g) * 0.35                  # 10 grazing rays
    d[134:144] = d[134:144] + tilt
    d = d / d.norm(dim=-1, keepdim=True)
    bf = synthetic_bitfield(N_GRID, 0.12, seed=2102)
    inputs = {'rays_o': o[None].contiguous(), 'rays_d': d[None].contiguous(), 'rays_r': torch.zeros(1, N, 1),
              'img': torch.rand(1, N, 3, generator=g), 'bkg_color': torch.rand(1, N, 3, generator=g)}
    return inputs, bf


def table_grad_summary(out, tag, grad, offsets):
    g = grad.astype(np.float32)
    L = len(offsets) - 1
    out[tag + 'low_levels'] = g[:offsets[3]].copy()
    rows = np.arange(5, g.shape[0], 16)
    out[tag + 'rows_mod16'] = g[rows].copy()
    out[tag + 'level_sum'] = np.stack([g[offsets[l]:offsets[l + 1]].astype(np.float64).sum(0) for l in range(L)])
    out[tag + 'level_abs'] = np.stack([np.abs(g[offsets[l]:offsets[l + 1]]).astype(np.float64).sum(0) for l in range(L)])
    sgn = np.random.default_rng(77).integers(0, 2, size=(4,) + g.shape, dtype=np.int8)
    out[tag + 'proj'] = np.array([((sgn[i].astype(np.float64) * 2 - 1) * g).sum() for i in range(4)])
    out[tag + 'max'] = np.array(np.abs(g).max())
    out[tag + 'nnz_rows'] = np.array(int((np.abs(g).sum(1) > 0).sum()))


def run(model, inputs, **kw):
    _state['samples'] = []
    return model({k: v.clone() for k, v in inputs.items()}, **kw)


def main():
    inputs, bf = make_inputs()
    out = {'bitfield_packed': np.packbits(bf.reshape(-1), bitorder='little'), 'table_seed': np.array(TABLE_SEED), 'table_amp': np.array(TABLE_AMP)}
    for k, v in inputs.items():
        out['in_' + k] = v.numpy()
    for bound in ('k2', 'tb'):
        ref_volume.aabb_ray_intersection = oracle_k2 if bound == 'k2' else _orig_aabb
        for net, ov in NETS.items():
            if bound == 'tb' and net != 'lin':
                continue
            tag = '{}_{}_'.format(bound, net)
            torch.manual_seed(2110)
            model = build_model(load_configs('/root/reference/configs/models/nerf_ngp.yaml', BASE + ov), None)
            fg = model.fg_model
            emb = fg.coarse_geo_net.embed_fn
            table = table_from_seed(emb.n_total_embed, emb.n_feat_per_entry)
            with torch.no_grad():
                emb.embeddings.copy_(torch.from_numpy(table))
                fg.obj_bound.volume.get_voxel_bitfield().copy_(torch.from_numpy(bf))
                last = fg.coarse_geo_net.layers[-1]
                last.weight[:1] *= 3.0   # density that matters: sigma = exp(out[0]) spans ~[0.05, 20]
            if 'table_sum' not in out:
                out['table_sum'] = np.array(table.astype(np.float64).sum())
                out['table_probe'] = table[::100003].copy()
                out['offsets'] = np.array(emb.offsets, np.int64)
                out['resolutions'] = np.array(emb.resolutions, np.int64)
            for k, v in model.state_dict().items():
                if not k.endswith(('embed_fn.embeddings', '.volume_pts', '.grid_pts', '.corner')) and 'bitfield' not in k and 'opafield' not in k:
                    out[tag + 'sd.' + k] = v.numpy()
            _state['rng'] = orc.Pcg32(9121)
            # 1. inference
            with torch.no_grad():
                res = run(model, inputs, inference_only=True)
            for k, v in res.items():
                out[tag + 'infer_' + k] = v.numpy()
            nr, fr, z, m = _state['samples'][0]
            out[tag + 'infer_near'], out[tag + 'infer_far'] = nr, fr
            out[tag + 'infer_zvals'], out[tag + 'infer_mask_pts'] = z[:, :max(2, int(m.sum(1).max()))], np.packbits(m, axis=1, bitorder='little')
            print(tag, 'infer: samples', int(m.sum()), 'max/ray', int(m.sum(1).max()), 'rays with samples', int((m.sum(1) > 0).sum()), 'of', m.shape[0])
            # 2. train, noise 0, with gradients
            fg.set_ray_cfgs('noise_std', 0.0)
            res = run(model, inputs, inference_only=False)
            loss = ((res['rgb_coarse'] - inputs['img']) ** 2).mean() * 100.0
            model.zero_grad()
            loss.backward()
            for k, v in res.items():
                if torch.is_tensor(v):
                    out[tag + 'train0_' + k] = v.detach().numpy()
            out[tag + 'train0_loss'] = loss.detach().numpy()
            nr, fr, z, m = _state['samples'][0]
            out[tag + 'train0_zvals'], out[tag + 'train0_mask_pts'] = z[:, :max(2, int(m.sum(1).max()))], np.packbits(m, axis=1, bitorder='little')
            for k, p in model.named_parameters():
                if p.grad is None:
                    continue
                if k.endswith('embed_fn.embeddings'):
                    table_grad_summary(out, tag + 'train0_tgrad_', p.grad.numpy(), emb.offsets)
                else:
                    out[tag + 'train0_grad.' + k] = p.grad.numpy().copy()
            out[tag + 'train0_dynamicbs_factor'] = np.array(float(model.get_dynamicbs_factor()))   # reading it resets the measurement
            print(tag, 'train0: loss', float(loss), 'table grad max', float(emb.embeddings.grad.abs().max()), 'dyn factor', out[tag + 'train0_dynamicbs_factor'])
            # 3. train, noise_std 1: record the reference's own draw
            fg.set_ray_cfgs('noise_std', 1.0)
            drawn = []
            real_randn = torch.randn

            def recording_randn(*a, **k):
                t = real_randn(*a, **k)
                drawn.append(t.clone())
                return t
            torch.manual_seed(2111)
            ref_rh.torch.randn = recording_randn
            try:
                res = run(model, inputs, inference_only=False)
            finally:
                ref_rh.torch.randn = real_randn
            assert len(drawn) == 1
            loss = ((res['rgb_coarse'] - inputs['img']) ** 2).mean() * 100.0
            model.zero_grad()
            loss.backward()
            for k, v in res.items():
                if torch.is_tensor(v):
                    out[tag + 'train1_' + k] = v.detach().numpy()
            out[tag + 'train1_loss'] = loss.detach().numpy()
            out[tag + 'train1_noise'] = drawn[0].numpy()     # (valid rays, P' or P'-1)
            nr, fr, z, m = _state['samples'][0]
            out[tag + 'train1_zvals'], out[tag + 'train1_mask_pts'] = z[:, :max(2, int(m.sum(1).max()))], np.packbits(m, axis=1, bitorder='little')
            for k, p in model.named_parameters():
                if p.grad is None:
                    continue
                if k.endswith('embed_fn.embeddings'):
                    table_grad_summary(out, tag + 'train1_tgrad_', p.grad.numpy(), emb.offsets)
                else:
                    out[tag + 'train1_grad.' + k] = p.grad.numpy().copy()
            print(tag, 'train1: loss', float(loss), 'noise', tuple(drawn[0].shape))
    out['overrides_base'] = np.array(BASE)
    for net, ov in NETS.items():
        out['overrides_' + net] = np.array(ov)
    path = os.path.join(OUT, 'g21_ngp_model.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) / 1024, 'KB')


if __name__ == '__main__':
    main()
