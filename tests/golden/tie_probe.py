"""Near-tie probe for the inverse-CDF resampling of the reference (build container only; imported by the generators).

`sample_cdf` (arcnerf/render/ray_helper.py:432-473) has two discontinuities in its inputs: the bin a uniform `u` falls into
(`searchsorted(cdf, u, right=True)`: a cdf edge one ulp either side of `u` moves the sample to the neighbouring bin) and the
`denom[denom < eps] = 1` rule (a bin whose cdf mass is one ulp either side of 1e-5 switches between `t = (u - c0) / denom` and
`t = u - c0`).  With `det=True` the lattice u = linspace(0, 1, n) contains u = 1.0 exactly, which is compared with cdf[-1] = a float
cumsum that is 1.0 to within an ulp: whether it rounds above 1.0 decides between bins[-1] and (for a flat last bin) ~bins[-2].  The
reference's own CPU and CUDA runs take different sides there; a parity fixture must not contain such rays.

`Probe` wraps the reference's sample_cdf, recomputes bins and denominators, and records per ray the smallest margin by which any of
the decisions could flip (in units of cdf mass)."""
import numpy as np
import torch


class Probe:
    def __init__(self, ray_helper_module):
        self.rh = ray_helper_module
        self.orig = ray_helper_module.sample_cdf
        self.margins = []      # one (n_rays,) array per call

    def __enter__(self):
        probe = self

        def wrapped(bins, cdf, n_sample, det=False, eps=1e-5):
            assert det, 'the probe assumes the deterministic lattice (perturb off)'
            u = torch.linspace(0.0, 1.0, steps=n_sample).expand(list(cdf.shape[:-1]) + [n_sample]).contiguous()
            c = cdf.detach()
            n_pts = c.shape[-1]
            inds = torch.searchsorted(c, u, right=True)
            below, above = torch.clamp(inds - 1, 0, n_pts - 1), torch.clamp(inds, 0, n_pts - 1)
            c0, c1 = torch.gather(c, 1, below), torch.gather(c, 1, above)
            b0, b1 = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
            width = (b1 - b0).abs()
            denom = c1 - c0
            # (a) u against EVERY cdf edge: moving across edge k changes the sample by about the local bin width unless both bins are
            #     linear pieces of the same slope; take the distance to the nearest edge, weighted by nothing (conservative)
            d_edge = (u[:, :, None] - c[:, None, :]).abs()
            # edges whose crossing cannot change the result: u == 0 against cdf[0] == 0 (clamped below), and crossings where the
            # sample position is continuous (both neighbouring bins regular: denom >= eps on both sides) - position is continuous in u
            # across a regular edge, so only edges next to a `denom < eps` bin or at the clamped ends count
            dc = c[:, 1:] - c[:, :-1]
            irregular = dc < eps * 1.02                      # bins that use (or nearly use) the denom = 1 rule
            edge_bad = torch.zeros_like(c, dtype=torch.bool)
            edge_bad[:, :-1] |= irregular
            edge_bad[:, 1:] |= irregular
            edge_bad[:, -1] = True                           # u = 1 against cdf[-1]: clamped end
            d_edge = torch.where(edge_bad[:, None, :].expand_as(d_edge), d_edge, torch.full_like(d_edge, 1.0))
            d_edge[:, 0, 0] = 1.0                            # u = 0 vs cdf[0] = 0: inds = 1 for any cdf[1] > 0
            m_edge = d_edge.amin(dim=(1, 2))
            # (b) the denom rule for the bins actually used, where the two forms differ by more than 1e-6 in position
            differs = ((u - c0) * (1.0 / denom.clamp_min(1e-12) - 1.0)).abs() * width > 1e-6
            m_den = torch.where(differs & (above != below), (denom - eps).abs(), torch.full_like(denom, 1.0)).amin(dim=1)
            probe.margins.append(torch.minimum(m_edge, m_den).numpy())
            return probe.orig(bins, cdf, n_sample, det, eps)
        self.rh.sample_cdf = wrapped
        return self

    def __exit__(self, *a):
        self.rh.sample_cdf = self.orig

    def per_ray(self):
        """smallest margin of each ray over all calls with that ray count"""
        by_n = {}
        for m in self.margins:
            by_n.setdefault(m.shape[0], []).append(m)
        return {n: np.min(np.stack(v), axis=0) for n, v in by_n.items()}
