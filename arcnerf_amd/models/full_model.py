"""FullModel (arcnerf/models/full_model.py:10-560): flattens (B, N, ...) ray batches, runs the foreground model in
`chunk_rays` chunks and reshapes back.  Background models (NeRF++ / MultiVol) are the next row of the scope table; a
config with `model.background` raises until that row is built."""
import torch
import torch.nn as nn

from ..utils.torch_utils import chunk_processing


class FullModel(nn.Module):
    def __init__(self, cfgs, fg_model, bkg_cfgs=None, bkg_model=None):
        super().__init__()
        if bkg_model is not None:
            raise NotImplementedError('background models are not on this path yet (SURVEY.md §8f row 2)')
        self.cfgs, self.fg_model, self.bkg_cfgs, self.bkg_model = cfgs, fg_model, None, None
        self.fg_only = False

    def get_fg_model(self):
        return self.fg_model

    def get_bkg_model(self):
        return None

    def get_chunk_rays(self):
        return self.fg_model.get_chunk_rays()

    def get_chunk_pts(self):
        return self.fg_model.get_chunk_pts()

    def set_chunk_rays(self, v):
        self.fg_model.set_chunk_rays(v)

    def set_chunk_pts(self, v):
        self.fg_model.set_chunk_pts(v)

    def init_setting(self):
        self.fg_model.init_setting()

    def is_cuda(self):
        return next(self.parameters()).is_cuda

    def sigma_reverse(self):
        return self.fg_model.sigma_reverse()

    def get_dynamicbs_factor(self):
        return self.fg_model.get_dynamicbs_factor()

    def reset_measurement(self):
        self.fg_model.reset_measurement()

    @staticmethod
    def clean_progress(output):
        for k in [k for k in output if k.startswith('progress_')]:
            output.pop(k)
        return output

    @staticmethod
    def detach_progress(output):
        for k in output:
            if k.startswith('progress_') and isinstance(output[k], torch.Tensor):
                output[k] = output[k].detach()
        return output

    def prepare_flatten_inputs(self, inputs):
        """(B, N, ...) -> (BN, ...) for img, rays_o, rays_d, rays_r and the optional bounds/mask/bkg_color/exp_time"""
        flat = {'img': inputs['img'].view(-1, 3) if 'img' in inputs else None,
                'rays_o': inputs['rays_o'].view(-1, 3), 'rays_d': inputs['rays_d'].view(-1, 3),
                'rays_r': inputs['rays_r'].view(-1, 1) if 'rays_r' in inputs else None,
                'bounds': inputs['bounds'].view(-1, 2) if 'bounds' in inputs else None,
                'mask': inputs['mask'].view(-1) if 'mask' in inputs else None,
                'bkg_color': inputs['bkg_color'].view(-1, 3) if 'bkg_color' in inputs else None,
                'exp_time': inputs['exp_time'].view(-1) if 'exp_time' in inputs else None}
        b, n = inputs['rays_o'].shape[:2]
        return flat, b, n

    @staticmethod
    def reshape_output(output, batch_size, n_rays_per_batch):
        for k, v in output.items():
            if isinstance(v, torch.Tensor) and v.shape[0] == batch_size * n_rays_per_batch:
                output[k] = v.view(batch_size, n_rays_per_batch, *v.shape[1:])
        return output

    def forward(self, inputs, inference_only=False, get_progress=False, cur_epoch=0, total_epoch=300000):
        flat, b, n = self.prepare_flatten_inputs(inputs)
        out = chunk_processing(self.process_fg_bkg_model, self.fg_model.get_chunk_rays(), False, self.fg_model, None, flat,
                               inference_only, get_progress, cur_epoch, total_epoch)
        return self.reshape_output(out, b, n)

    def process_fg_bkg_model(self, fg_model, bkg_model, flat_inputs, inference_only, get_progress, cur_epoch, total_epoch):
        out = fg_model.forward(flat_inputs, inference_only, get_progress, cur_epoch, total_epoch)
        if not get_progress:
            out = self.clean_progress(out)
        return self.detach_progress(out)

    @torch.no_grad()
    def optimize(self, cur_epoch=0):
        self.fg_model.optimize(cur_epoch)

    def forward_pts_dir(self, pts, view_dir=None):
        return self.fg_model.forward_pts_dir(pts, view_dir)

    def forward_pts(self, pts):
        return self.fg_model.forward_pts(pts)
