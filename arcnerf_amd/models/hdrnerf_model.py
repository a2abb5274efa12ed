"""HDRNeRF (arcnerf/models/hdrnerf_model.py:12-171): NeRF whose radiance is a log-HDR value; three tiny 1 -> W -> 1 tone-mapping
MLPs (one per colour channel) turn rgb_h + ln(exposure time) into the LDR colour that is composited and supervised, a second
compositing pass of exp(rgb_h) gives the HDR image, and `unit_exp` (the tone-mapped value of rgb_h = 0, dt = 1) feeds the
unit-exposure constraint.  SURVEY.md section 8f, rank 3.  Both compositing passes run the ray-marching kernel; the tone mappers
are nn.Linear stacks as in the reference."""
import torch
import torch.nn as nn

from ..utils.cfgs_utils import dict_to_obj, get_value_from_cfgs_field
from ..utils.registry import MODEL_REGISTRY
from .base_modules.activation import get_activation
from .base_modules.linear import DenseLayer
from .nerf_model import NeRF



def _fused_tone_mappers(mlps, x):
    """the three 1 -> W -> 1 tone mappers (ReLU inside, sigmoid out, biases, W <= 128) in ONE kernel per direction with the hidden layer
    in registers (ops.autograd.ToneMapFn): as dense layers each channel moved a (samples, W) tensor through HBM four times per step.
    Same parameters (the layers' own weight / bias, packed per call: 3 x 385 floats).  None where that shape does not apply."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.numel() > 0):
        return None
    rows = []
    for layers in mlps:
        if len(layers) != 2 or any(type(la) is not DenseLayer or hasattr(la, 'weight_g') or la.bias is None for la in layers):
            return None
        if type(layers[0].activation) is not nn.ReLU or type(layers[1].activation) is not nn.Sigmoid:
            return None
        if layers[0].in_features != 1 or layers[1].out_features != 1 or layers[0].out_features > 128:
            return None
        rows.append(torch.cat([layers[0].weight.reshape(-1), layers[0].bias.reshape(-1), layers[1].weight.reshape(-1), layers[1].bias.reshape(-1)]))
    if len({r.numel() for r in rows}) != 1:
        return None
    from ..ops.autograd import ToneMapFn
    return ToneMapFn.apply(x.contiguous(), torch.stack(rows))

@MODEL_REGISTRY.register()
class HDRNeRF(NeRF):
    def __init__(self, cfgs):
        super().__init__(cfgs)
        exp_cfgs = self.cfgs.model.exp_mlps
        self.coarse_exp_r_mlps, self.coarse_exp_g_mlps, self.coarse_exp_b_mlps = self.build_exp_mlps(exp_cfgs)
        if self.get_ray_cfgs('n_importance') > 0:
            if self.get_ray_cfgs('shared_network'):
                self.fine_exp_r_mlps, self.fine_exp_g_mlps, self.fine_exp_b_mlps = \
                    self.coarse_exp_r_mlps, self.coarse_exp_g_mlps, self.coarse_exp_b_mlps
            else:
                self.fine_exp_r_mlps, self.fine_exp_g_mlps, self.fine_exp_b_mlps = self.build_exp_mlps(exp_cfgs)

    def packed_path_eligible(self):
        return False  # the packed NGP step has no tone-mapping stage

    def build_exp_mlps(self, cfgs):
        act_cfgs = get_value_from_cfgs_field(cfgs, 'act_cfgs', None)
        out_act_cfgs = get_value_from_cfgs_field(cfgs, 'out_act_cfg', None)
        sep = []
        for _ in range(3):
            layers = []
            for i in range(cfgs.D + 1):
                in_dim, out_dim = (1 if i == 0 else cfgs.W), (1 if i == cfgs.D else cfgs.W)
                if i != cfgs.D:
                    layers.append(DenseLayer(in_dim, out_dim, activation=get_activation(act_cfgs)))
                else:
                    layers.append(DenseLayer(in_dim, out_dim,
                                             activation=get_activation(out_act_cfgs, dict_to_obj({'type': 'Sigmoid'}))))
            sep.append(nn.ModuleList(layers))
        return sep[0], sep[1], sep[2]

    @staticmethod
    def forward_exp_mlps(l_r, l_g, l_b, rgb_h, exp_time):
        """rgb_h (B,3) log-HDR, exp_time (B,) -> LDR rgb (B,3)"""
        log_t = torch.log(exp_time)
        fused = _fused_tone_mappers((l_r, l_g, l_b), rgb_h + log_t[:, None])
        if fused is not None:
            return fused
        chans = []
        for c, layers in enumerate((l_r, l_g, l_b)):
            h = (rgb_h[:, c] + log_t)[:, None]
            for layer in layers:
                h = layer(h)
            chans.append(h)
        return torch.cat(chans, -1)

    def _stage(self, nets, mlps, rays_o, rays_d, zvals, mask_pts, bkg_color, exp_time, inference_only):
        sigma, rgb_h = self.get_sigma_radiance_by_mask_pts(nets[0], nets[1], rays_o, rays_d, zvals, mask_pts, inference_only)
        exp_rep = torch.repeat_interleave(exp_time, rgb_h.shape[1], 0)
        rgb_l = self.forward_exp_mlps(*mlps, rgb_h.view(-1, 3), exp_rep).view(rays_o.shape[0], -1, 3)
        out = self.ray_marching(sigma, rgb_l, zvals, inference_only=inference_only, bkg_color=bkg_color)
        if 'rgb' in out:
            out['hdr'] = self.ray_marching(sigma, torch.exp(rgb_h), zvals, inference_only=inference_only, bkg_color=bkg_color)['rgb']
        if not inference_only:
            out['unit_exp'] = self.point_constraint(*mlps)
        return out

    def _forward(self, inputs, inference_only=False, get_progress=False, cur_epoch=0, total_epoch=300000):
        rays_o, rays_d, zvals = inputs['rays_o'], inputs['rays_d'], inputs['zvals']
        mask_pts, bkg_color, exp_time = inputs['mask_pts'], inputs['bkg_color'], inputs['exp_time']
        output = {}
        out_c = self._stage((self.coarse_geo_net, self.coarse_radiance_net),
                            (self.coarse_exp_r_mlps, self.coarse_exp_g_mlps, self.coarse_exp_b_mlps), rays_o, rays_d, zvals,
                            mask_pts, bkg_color, exp_time, inference_only)
        weights_c = out_c['weights']
        output['coarse'] = self.output_get_progress(out_c, get_progress)
        if self.get_ray_cfgs('n_importance') > 0:
            zvals, mask_pts = self.upsample_zvals(zvals, weights_c, mask_pts, inference_only)
            out_f = self._stage((self.fine_geo_net, self.fine_radiance_net),
                                (self.fine_exp_r_mlps, self.fine_exp_g_mlps, self.fine_exp_b_mlps), rays_o, rays_d, zvals, mask_pts,
                                bkg_color, exp_time, inference_only)
            output['fine'] = self.output_get_progress(out_f, get_progress)
        return self.adjust_coarse_fine_output(output, inference_only)

    def point_constraint(self, l_r, l_g, l_b):
        """tone-mapped value of rgb_h = 0 at unit exposure (log 1 = 0)"""
        device = next(self.coarse_exp_b_mlps.parameters()).device
        return self.forward_exp_mlps(l_r, l_g, l_b, torch.zeros([1, 3], device=device), torch.ones([1], device=device))
