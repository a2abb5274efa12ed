"""Evaluate the nets on the VALID samples of a padded (rays, slots) layout only.

The pruned samplers return zvals (B, P) with a [T..T F..F] mask per ray, padded slots repeating the last valid z.  The
reference gathers the valid points, runs the nets once, and fills every padded slot with its ray's last valid value so that the
compositor sees zero-length intervals there (fg_model.py:331-386, multivol_bkg_model.py:150-196); rays without any sample are
split off beforehand and get sigma = 0.  Here that is one function, and the empty rays are handled in place (no second gather)."""
import torch

from ..geometry.ray import get_ray_points_by_zvals
from ..utils.torch_utils import chunk_processing


def nets_on_valid_samples(forward_pts_dir, chunk_pts, geo_net, radiance_net, rays_o, rays_d, zvals, mask_pts):
    """-> sigma (B, P), radiance (B, P, 3).  Rows of `mask_pts` without a True get zeros."""
    n_rays, n_pts = zvals.shape
    counts = mask_pts.sum(dim=1)
    sigma = zvals.new_zeros((n_rays, n_pts))
    radiance = zvals.new_zeros((n_rays, n_pts, 3))
    flat = mask_pts.reshape(-1).nonzero(as_tuple=True)[0]   # one nonzero (one host sync) for both gathers and both scatters
    if flat.numel() == 0:
        return sigma, radiance
    pts = get_ray_points_by_zvals(rays_o, rays_d, zvals).reshape(-1, 3).index_select(0, flat)
    dirs = rays_d.index_select(0, torch.div(flat, n_pts, rounding_mode='floor'))
    s_valid, r_valid = chunk_processing(forward_pts_dir, chunk_pts, False, geo_net, radiance_net, pts.contiguous(), dirs.contiguous())
    has = counts > 0
    last = (torch.cumsum(counts, dim=0) - 1).clamp_min(0)   # flat index of each ray's last valid sample
    sigma = torch.where(has[:, None], s_valid[last][:, None].expand(n_rays, n_pts), sigma).contiguous()
    radiance = torch.where(has[:, None, None], r_valid[last][:, None, :].expand(n_rays, n_pts, 3), radiance).contiguous()
    sigma = sigma.view(-1).index_copy(0, flat, s_valid.reshape(-1)).view(n_rays, n_pts)
    radiance = radiance.view(-1, 3).index_copy(0, flat, r_valid.reshape(-1, 3)).view(n_rays, n_pts, 3)
    return sigma, radiance


def hold_rays(handle, rays_o, rays_d):
    """A sampler handle that may be picked up by a LATER forward is matched by the rays' (pointer, size, version): it keeps the ray tensors
    alive - the caching allocator cannot hand their addresses to another batch while the entry exists - and tells the allocator that the
    stream the sampler was queued on (the sampling stream, FullModel.prefetch_samples) reads them."""
    handle['rays'] = (rays_o, rays_d)
    if rays_o.is_cuda:
        cur = torch.cuda.current_stream(rays_o.device)
        rays_o.record_stream(cur)
        rays_d.record_stream(cur)
