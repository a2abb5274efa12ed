"""ORACLE — TEST INFRASTRUCTURE ONLY: BASELINE config 1 (vanilla NeRF, configs/models/nerf.yaml) as plain PyTorch-CPU eager code.

The like-for-like stand-in for the reference's CPU path `scripts/cpu.sh` -> `train.py --gpu_ids -1 --configs configs/default.yaml`
(NeRF, frequency encoders, 64 + 128 samples per ray, n_rays 4096, Adam): the same op sequence as the reference modules, written
out flat so that it can run on the GPU box's host cores where /root/reference does not exist.  Used by bench.py's `cpu_baseline`
leg only; pinned to a run of the reference itself by tests/test_oracle_golden.py (golden G22, the reference FullModel at full width).

Follows: FgModel.forward / get_near_far (ray_helper.py:181-228, hard-coded near/far), get_zvals_from_near_far (:231-265),
get_ray_points_by_zvals (geometry/ray.py:11-30), FreqEmbedder.forward (encoding/freq_encoder.py:65-88), GeoNet.forward
(linear_network_module.py:174-197, skip concat [h, x_embed]), RadianceNet.forward + fuse_radiance_inputs mode 'vf'
(:318-335, encoder_mlp_network.py:93-118, re-normalised view dirs), ray_marching (ray_helper.py:476-593), NeRF.upsample_zvals
(nerf_model.py:93-117) with sample_pdf / sample_cdf (ray_helper.py:410-473).
"""
import math

import torch
import torch.nn as nn


def freq_embed(x, n_freqs):
    """[x, sin(2^k x), cos(2^k x)]_k (include_input, log-sampled bands 2^0 .. 2^(n-1))"""
    out = [x]
    for k in range(n_freqs):
        f = 2.0 ** k
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


class _Net(nn.Module):
    """one geometry + radiance pair with the reference's parameter names (layers.i.weight / .bias)"""

    def __init__(self, W=256, D=8, skips=(4,), W_feat=256, n_freqs=10, W_rad=128, n_freqs_view=4):
        super().__init__()
        self.D, self.skips, self.n_freqs, self.n_freqs_view = D, tuple(skips), n_freqs, n_freqs_view
        e = 3 + 6 * n_freqs
        ev = 3 + 6 * n_freqs_view
        geo = []
        for i in range(D + 1):
            in_dim = e if i == 0 else (W + e if (i - 1) in self.skips else W)
            geo.append(nn.Linear(in_dim, (1 + W_feat) if i == D else W))
        rad = [nn.Linear(ev + W_feat, W_rad), nn.Linear(W_rad, 3)]
        self.geo, self.rad = nn.ModuleList(geo), nn.ModuleList(rad)

    def forward(self, pts, dirs):
        xe = freq_embed(pts, self.n_freqs)
        h = xe
        for i in range(self.D + 1):
            h = self.geo[i](h)
            if i < self.D:
                h = torch.relu(h)
            if i in self.skips:
                h = torch.cat([h, xe], -1)
        sigma, feat = h[:, 0], h[:, 1:]
        d = dirs / (dirs.norm(dim=-1, keepdim=True) + 1e-8)      # encoder_mlp_network.py:100
        r = torch.cat([freq_embed(d, self.n_freqs_view), feat], -1)   # mode 'vf'
        rgb = torch.sigmoid(self.rad[1](torch.relu(self.rad[0](r))))
        return sigma, rgb


def ray_marching(sigma, radiance, zvals, bkg_color=None, noise_std=0.0):
    """add_inf_z = True branch of ray_helper.ray_marching"""
    deltas = zvals[:, 1:] - zvals[:, :-1]
    deltas = torch.where(deltas.abs() < 1e-5, torch.zeros_like(deltas), deltas)
    deltas = torch.cat([deltas, 1e10 * torch.ones_like(deltas[:, :1])], -1)
    noise = torch.randn_like(sigma) * noise_std if noise_std > 0 else 0.0
    alpha = 1 - torch.exp(-torch.relu(sigma + noise) * deltas)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], -1), -1)[:, :-1]
    w = alpha * trans
    rgb = (w[..., None] * radiance).sum(-2)
    if bkg_color is not None:
        rgb = rgb + trans[:, -1:] * bkg_color
    return rgb, (w * zvals).sum(-1), w.sum(-1), w


def sample_pdf(bins, weights, n_sample, det=True, eps=1e-5):
    weights = weights + eps
    pdf = weights / weights.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    n_pts = bins.shape[-1]
    if det:
        u = torch.linspace(0.0, 1.0, steps=n_sample).expand(cdf.shape[0], n_sample).contiguous()
    else:
        u = torch.rand(cdf.shape[0], n_sample)
    inds = torch.searchsorted(cdf.detach(), u, right=True)
    below, above = torch.clamp(inds - 1, 0, n_pts - 1), torch.clamp(inds, 0, n_pts - 1)
    c0, c1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    b0, b1 = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = c1 - c0
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)
    samples = b0 + (u - c0) / denom * (b1 - b0)
    return torch.sort(samples, -1)[0]


class TorchCpuNerf(nn.Module):
    """coarse + fine nets, 64 + 128 samples; parameter names map onto the reference state_dict through `load_reference_state`"""

    def __init__(self, n_sample=64, n_importance=128, near=1.5, far=8.0):
        super().__init__()
        self.coarse, self.fine = _Net(), _Net()
        self.n_sample, self.n_importance, self.near, self.far = n_sample, n_importance, near, far

    def load_reference_state(self, sd):
        """sd: {reference key: tensor} (fg_model.{coarse,fine}_{geo,radiance}_net.layers.i.{weight,bias})"""
        own = {}
        for tag, net in (('coarse', self.coarse), ('fine', self.fine)):
            for kind, mods in (('geo', net.geo), ('radiance', net.rad)):
                for i, m in enumerate(mods):
                    for p in ('weight', 'bias'):
                        getattr(m, p).data.copy_(sd['fg_model.{}_{}_net.layers.{}.{}'.format(tag, kind, i, p)])
        return own

    def _render(self, net, o, d, z, bkg, noise_std):
        R, P = z.shape
        pts = (o[:, None, :] + z[..., None] * d[:, None, :]).reshape(-1, 3)
        dirs = d[:, None, :].expand(R, P, 3).reshape(-1, 3)
        sigma, rgb = net(pts, dirs)
        return ray_marching(sigma.view(R, P), rgb.view(R, P, 3), z, bkg, noise_std)

    def forward(self, o, d, bkg=None, perturb=False, noise_std=0.0):
        R = o.shape[0]
        t = torch.linspace(0.0, 1.0, self.n_sample)
        z = (self.near + (self.far - self.near) * t)[None].expand(R, -1).contiguous()
        if perturb:   # perturb_interval: uniform inside the mid-point intervals
            mids = 0.5 * (z[:, 1:] + z[:, :-1])
            upper, lower = torch.cat([mids, z[:, -1:]], -1), torch.cat([z[:, :1], mids], -1)
            z = lower + (upper - lower) * torch.rand_like(z)
        rgb_c, depth_c, mask_c, w = self._render(self.coarse, o, d, z, bkg, noise_std)
        mids = 0.5 * (z[:, 1:] + z[:, :-1])
        new = sample_pdf(mids, w[:, 1:self.n_sample - 1].detach(), self.n_importance, det=not perturb).detach()
        z_f = torch.sort(torch.cat([z, new], -1), -1)[0]
        rgb_f, depth_f, mask_f, _ = self._render(self.fine, o, d, z_f, bkg, noise_std)
        return {'rgb_coarse': rgb_c, 'depth_coarse': depth_c, 'mask_coarse': mask_c, 'rgb_fine': rgb_f, 'depth_fine': depth_f,
                'mask_fine': mask_f}


def time_train_steps(n_rays, steps=1, threads=None, seed=0):
    """fwd + bwd + Adam of config 1 for n_rays rays on the host cores; returns (net evaluations per step, seconds per step)"""
    import time
    if threads:
        torch.set_num_threads(int(threads))
    torch.manual_seed(seed)
    m = TorchCpuNerf()
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)
    g = torch.Generator().manual_seed(seed + 1)
    o = torch.randn(n_rays, 3, generator=g)
    o = o / o.norm(dim=-1, keepdim=True) * 4.0
    d = -o + (torch.rand(n_rays, 3, generator=g) - 0.5) * 1.5
    d = d / d.norm(dim=-1, keepdim=True)
    img, bkg = torch.rand(n_rays, 3, generator=g), torch.rand(n_rays, 3, generator=g)

    def step():
        out = m(o, d, bkg, perturb=True, noise_std=1.0)
        loss = ((out['rgb_fine'] - img) ** 2).mean() + ((out['rgb_coarse'] - img) ** 2).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    step()   # warm-up (allocator, thread pool)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    return n_rays * (m.n_sample + m.n_sample + m.n_importance), dt
