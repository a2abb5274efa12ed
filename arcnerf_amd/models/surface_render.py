"""Surface rendering (inference only): one ray / surface intersection per ray, then the radiance net at that point.

One implementation behind Base3dModel.surface_render (base_3d_model.py:307-366), FgModel.surface_render (fg_model.py:412-470: rays
that miss the object bound are skipped) and SdfModel.surface_render (sdf_model.py:116-185: also returns the normal)."""
import torch

from ..geometry.surface import surface_ray_intersection


def render_surface(model, inputs, method, n_step, n_iter, threshold, level, grad_dir, with_normal=False):
    """-> {'rgb' (B,3) white where nothing is hit, 'depth' (B,1) the intersection zvals, 'mask' (B,) float[, 'normal' (B,3)]}"""
    rays_o, rays_d = inputs['rays_o'], inputs['rays_d']
    n_rays = rays_o.shape[0]
    near, far, hits_bound = model.get_near_far_from_rays(inputs)
    geo_net, radiance_net = model.get_net()
    if hits_bound is None or bool(torch.all(hits_bound)):
        zvals, pts, mask = surface_ray_intersection(rays_o, rays_d, geo_net.forward_geo_value, method, near, far, n_step, n_iter,
                                                    threshold, level, grad_dir)
    else:
        z_in, p_in, m_in = surface_ray_intersection(rays_o[hits_bound], rays_d[hits_bound], geo_net.forward_geo_value, method,
                                                    near[hits_bound], far[hits_bound], n_step, n_iter, threshold, level, grad_dir)
        # rays outside the bound: depth = the largest depth found, no hit
        zvals = torch.ones((n_rays, 1), dtype=rays_o.dtype, device=rays_o.device) * z_in.max()
        pts = torch.ones((n_rays, 3), dtype=rays_o.dtype, device=rays_o.device)
        mask = torch.zeros((n_rays,), dtype=torch.bool, device=rays_o.device)
        zvals[hits_bound], pts[hits_bound], mask[hits_bound] = z_in, p_in, m_in
    out = {'rgb': torch.ones((n_rays, 3), dtype=rays_o.dtype, device=rays_o.device), 'depth': zvals, 'mask': mask.type(rays_o.dtype)}
    if with_normal:
        out['normal'] = torch.zeros((n_rays, 3), dtype=rays_o.dtype, device=rays_o.device)
    if bool(torch.any(mask)):
        shaded = model._forward_pts_dir(geo_net, radiance_net, pts[mask], rays_d[mask])
        out['rgb'][mask] = shaded[1].detach()
        if with_normal:
            out['normal'][mask] = shaded[2].detach()
    return out
