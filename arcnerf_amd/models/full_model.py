"""FullModel (arcnerf/models/full_model.py:10-560): flattens (B, N, ...) ray batches, runs the foreground model (and the
background model, when `model.background` is configured) in `chunk_rays` chunks, blends the two and reshapes back.

Blending (full_model.py:141-349): `rgb` mode adds the background colour / depth scaled by the transmittance left after the
foreground's last sample; `sigma` mode concatenates the per-sample densities and colours of both models along the ray and
composites them again in one pass.  The foreground mask stays the foreground's."""

import torch
import torch.nn as nn

from ..utils.cfgs_utils import get_value_from_cfgs_field
from ..utils.torch_utils import chunk_processing


def _stage_keys(output):
    """suffixes under which a model reported its results: [''] (one stage) or ['_coarse'(, '_fine')]"""
    if any(k.endswith('_coarse') or k.endswith('_fine') for k in output):
        return ['_coarse'] + (['_fine'] if any(k.endswith('_fine') for k in output) else [])
    return ['']


class FullModel(nn.Module):
    def __init__(self, cfgs, fg_model, bkg_cfgs=None, bkg_model=None):
        super().__init__()
        self.cfgs, self.fg_model, self.bkg_cfgs, self.bkg_model = cfgs, fg_model, bkg_cfgs, bkg_model
        self.fg_only = False
        if self.bkg_cfgs is not None:
            self.bkg_blend = get_value_from_cfgs_field(self.bkg_cfgs.model, 'bkg_blend', 'rgb')
            self.check_bkg_cfgs()
            if self.bkg_blend == 'sigma':
                self.fg_model.set_add_inf_z(True)  # keep every foreground sigma for the joint compositing
            self.fg_only = get_value_from_cfgs_field(self.bkg_cfgs.model, 'fg_only', False)

    def check_bkg_cfgs(self):
        if self.bkg_blend == 'rgb':
            assert self.fg_model.get_ray_cfgs('add_inf_z') is False, 'Do not add_inf_z for foreground'
            assert self.bkg_model.get_ray_cfgs('add_inf_z') is True, 'Must use add_inf_z for background in rgb blending mode'
        elif self.bkg_blend == 'sigma':
            assert self.bkg_model.get_ray_cfgs('add_inf_z') is False, 'Do not add_inf_z for background in sigma blending mode'
        else:
            raise NotImplementedError('Invalid bkg_blend type {}'.format(self.bkg_blend))

    def get_fg_model(self):
        return self.fg_model

    def get_bkg_model(self):
        return self.bkg_model

    def get_chunk_rays(self):
        return self.fg_model.get_chunk_rays()

    def get_chunk_pts(self):
        return self.fg_model.get_chunk_pts()

    def set_chunk_rays(self, v):
        self.fg_model.set_chunk_rays(v)
        if self.bkg_model is not None:
            self.bkg_model.set_chunk_rays(v)

    def set_chunk_pts(self, v):
        self.fg_model.set_chunk_pts(v)
        if self.bkg_model is not None:
            self.bkg_model.set_chunk_pts(v)

    def init_setting(self):
        self.fg_model.init_setting()
        if self.bkg_model is not None:
            self.bkg_model.init_setting()

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

    @staticmethod
    def clean_two_stage_progress(output):
        """keep ONE set of progress_ entries: the one-stage set if present, else the fine one, else the coarse one, un-suffixed"""
        prog = [k for k in output if k.startswith('progress_')]
        if not prog:
            return output
        staged = [k for k in prog if k.endswith('_coarse') or k.endswith('_fine')]
        if len(staged) < len(prog):
            for k in staged:
                output.pop(k)
            return output
        keep = '_fine' if any(k.endswith('_fine') for k in prog) else '_coarse'
        for k in staged:
            v = output.pop(k)
            if k.endswith(keep):
                output[k[:-len(keep)]] = v
        return output

    def blend_bkg_rgb(self, fg_output, bkg_output):
        """rgb, depth += T_fg(after the last sample) * background (full_model.py:278-330).  A two-stage foreground takes the
        background stage of the same name when there is one, else the background's coarse / only stage."""
        fg_stages, bkg_stages = _stage_keys(fg_output), _stage_keys(bkg_output)
        for st in fg_stages:
            assert 'progress_trans_shift' + st in fg_output, 'You must get_progress for fg_model'
            if st in bkg_stages:
                bst = st
            elif st == '':
                bst = bkg_stages[-1]       # one-stage foreground: the finest background available
            else:
                bst = bkg_stages[0]        # '_coarse' or '' of the background
            lam = fg_output['progress_trans_shift' + st][:, -1]
            fg_output['rgb' + st] = fg_output['rgb' + st] + lam[:, None] * bkg_output['rgb' + bst]
            fg_output['depth' + st] = fg_output['depth' + st] + lam * bkg_output['depth' + bst]
        return self.clean_two_stage_progress(fg_output) if fg_stages != [''] else fg_output

    def blend_bkg_sigma(self, fg_output, bkg_output, inference_only=False, get_progress=False):
        """joint compositing of the foreground's and the background's samples (full_model.py:141-276)"""
        fg_stages, bkg_stages = _stage_keys(fg_output), _stage_keys(bkg_output)
        out = {}
        for st in fg_stages:
            assert 'progress_sigma' + st in fg_output, 'You must get_progress for fg_model'
            bst = st if st in bkg_stages else (bkg_stages[-1] if st == '' else bkg_stages[0])
            z_fg, z_bkg = fg_output['progress_zvals' + st], bkg_output['progress_zvals' + bst]
            s_fg, r_fg = fg_output['progress_sigma' + st], fg_output['progress_radiance' + st]
            # rays whose foreground samples run past the first background shell contribute nothing from the foreground
            invalid = z_fg[:, -1] > z_bkg[:, 0]
            s_fg[invalid], r_fg[invalid], z_fg[invalid] = 0, 0, 0
            joint = self.fg_model.ray_marching(torch.cat([s_fg, bkg_output['progress_sigma' + bst]], 1),
                                               torch.cat([r_fg, bkg_output['progress_radiance' + bst]], 1),
                                               torch.cat([z_fg, z_bkg], 1), inference_only=inference_only)
            joint = self.fg_model.output_get_progress(joint, get_progress, s_fg.shape[1])
            for k, v in joint.items():
                out[k + st] = fg_output[k + st] if (k == 'mask' and fg_stages != [''] and k + st in fg_output) else v
        return self.clean_two_stage_progress(out) if fg_stages != [''] else out

    def blend_output(self, fg_output, bkg_output=None, inference_only=False, get_progress=False):
        if bkg_output is None:
            final = self.clean_two_stage_progress(fg_output)
        elif self.bkg_blend == 'rgb':
            final = self.blend_bkg_rgb(fg_output, bkg_output)
        elif self.bkg_blend == 'sigma':
            final = self.blend_bkg_sigma(fg_output, bkg_output, inference_only, get_progress)
        else:
            raise NotImplementedError('Invalid bkg_blend type {}...'.format(self.bkg_blend))
        return final if get_progress else self.clean_progress(final)

    def forward(self, inputs, inference_only=False, get_progress=False, cur_epoch=0, total_epoch=300000):
        flat, b, n = self.prepare_flatten_inputs(inputs)
        chunk_rays = self.fg_model.get_chunk_rays()
        if self.bkg_model is not None:
            chunk_rays = min(chunk_rays, self.bkg_model.get_chunk_rays())
        out = chunk_processing(self.process_fg_bkg_model, chunk_rays, False, self.fg_model, self.bkg_model, flat,
                               inference_only, get_progress, cur_epoch, total_epoch)
        return self.reshape_output(out, b, n)

    @torch.no_grad()
    def prefetch_samples(self, inputs):
        """Queue the SAMPLING of a later forward now, on a second stream: `inputs` = the (B, N, ...) batch of the NEXT training step (the
        same tensors must be passed to that forward).  Both samplers - the foreground's occupancy marcher and the background's cascade
        marcher - with their scans run while the current step's backward and optimiser occupy the main stream, and their sample totals
        (the host reads that size the packed tensors) have arrived when the next forward asks for them: no marcher and no host wait on
        the step's critical path (config 4: 0.30 ms of kernels and two pipeline stalls per step).  The reference's trainer knows its
        next batch just as well (Pipeline.get_train_batch slices a pre-shuffled tensor); its samplers run inside forward.  Samplers that
        have no packed path, batches larger than one chunk, `sigma` blending and CPU tensors are left to the forward."""
        fg, bkg = self.fg_model, self.bkg_model
        if not inputs['rays_o'].is_cuda:
            return False
        flat, _, _ = self.prepare_flatten_inputs(inputs)
        n = flat['rays_o'].shape[0]
        chunk = fg.get_chunk_rays() if bkg is None else min(fg.get_chunk_rays(), bkg.get_chunk_rays())
        if chunk is not None and 0 < chunk < n:
            return False
        todo = [mdl for mdl in (fg, bkg if (bkg is not None and not self.fg_only and self.bkg_blend != 'sigma') else None)
                if mdl is not None and hasattr(mdl, 'presample')]
        if not todo:
            return False
        if getattr(self, '_sample_stream', None) is None or self._sample_stream.device != flat['rays_o'].device:
            self._sample_stream = torch.cuda.Stream(device=flat['rays_o'].device)
        main = torch.cuda.current_stream()
        self._sample_stream.wait_stream(main)       # the occupancy the samplers read, the rays: whatever the main stream wrote so far
        with torch.cuda.stream(self._sample_stream):
            for mdl in todo:
                mdl.presample(flat)
        return True

    def process_fg_bkg_model(self, fg_model, bkg_model, flat_inputs, inference_only, get_progress, cur_epoch, total_epoch):
        get_progress_fg = True if bkg_model is not None else get_progress   # blending needs the foreground's progress
        if bkg_model is not None and not get_progress and self.bkg_blend == 'rgb' and not self.fg_only:
            # ... of which `rgb` blending reads one number per ray, trans_shift[:, -1] (blend_bkg_rgb): a foreground with a packed path
            # may return just that (truthy, so every other model gives its full progress as before)
            get_progress_fg = 't_last'
        if bkg_model is not None and not self.fg_only and self.bkg_blend != 'sigma' and hasattr(bkg_model, 'presample'):
            if getattr(bkg_model, '_presampled', None) is None or bkg_model._presampled[0][:3] != (flat_inputs['rays_o'].data_ptr(), flat_inputs['rays_d'].data_ptr(), flat_inputs['rays_o'].shape[0]):
                bkg_model.presample(flat_inputs)   # its sampler runs (and its sample count travels) while the foreground works
        fg_output = fg_model.forward(flat_inputs, inference_only, get_progress_fg, cur_epoch, total_epoch)
        bkg_output = None
        if bkg_model is not None and not self.fg_only:
            # the background's per-sample progress is only consumed by the joint compositing of the `sigma` blend (the reference asks
            # for it unconditionally, full_model.py:437): in `rgb` mode a background may take its packed path
            bkg_output = bkg_model.forward(flat_inputs, inference_only, self.bkg_blend == 'sigma', cur_epoch, total_epoch)
        return self.detach_progress(self.blend_output(fg_output, bkg_output, inference_only, get_progress))

    @torch.no_grad()
    def optimize(self, cur_epoch=0):
        self.fg_model.optimize(cur_epoch)
        if self.bkg_model is not None:
            self.bkg_model.optimize(cur_epoch)

    def surface_render(self, inputs, method='sphere_tracing', n_step=128, n_iter=100, threshold=0.01, level=0.0, grad_dir='ascent',
                       **kwargs):
        """inference only, foreground model only: inputs (B, N, ...) -> rgb (B, N, 3), depth, mask[, normal] (full_model.py:477-524)"""
        flat, batch_size, n_rays_per_batch = self.prepare_flatten_inputs(inputs)
        out = chunk_processing(self.fg_model.surface_render, self.fg_model.get_chunk_rays(), False, flat, method, n_step, n_iter,
                               threshold, level, grad_dir)
        return self.reshape_output(out, batch_size, n_rays_per_batch)

    def forward_pts_dir(self, pts, view_dir=None):
        return self.fg_model.forward_pts_dir(pts, view_dir)

    def forward_pts(self, pts):
        return self.fg_model.forward_pts(pts)
