"""Pipeline (arcnerf/trainer/pipeline.py:9-317): the training rays of an iteration - centre precrop, shuffle, dynamic batch size, the batch
and the random background colour blended into its target.

Same class, same methods, same state machine (`train_sample_info`, `crop_max_epoch`, `init_precrop`) and the same order of random draws as
the reference; what differs is where the data lives.  The reference CROPS the concatenated per-pixel tensors (`step_crop_center_image`),
GATHERS all of them through a randperm (`step_ray_sample`: img, mask, rays_o, rays_d, rays_r = 44 B per training pixel, rewritten at every
shuffle) and slices a batch.  Here the dataset tensors are never rewritten: the crop is a window, the shuffle a permutation of int64 ray
ids on the GPU, and a batch is ONE launch (`arcn_fetch_train_batch`, csrc/batch.hip) that turns `ids[count : count + n_rays]` into pixels,
their rays through the view's camera (when the dataset carries `intrinsic` / `c2w` instead of precomputed rays), colours, masks and the
blended targets.

Datasets (`train_data`, what arcnerf_trainer.py:188-219 concat_train_batch collects): a dict of tensors in (n_img, H*W, ...) plus 'H', 'W':
    {'img' (N,HW,3), 'mask' (N,HW)}  or  {'rgba' (N,HW,4) uint8 as the PNGs hold them}                 colours
    {'rays_o', 'rays_d' (N,HW,3), 'rays_r' (N,HW,1)}  or  {'intrinsic' (N,3,3), 'c2w' (N,4,4)}         rays: precomputed, or cameras
    any other (N,HW,...) tensor ('bounds', 'exp_time', ...): gathered by row
`center_pixel` / `normalize_rays_d` (dataset cfgs of the reference, default True / True) may ride in the dict for the camera form.
"""
import math

import torch

from ..ops import functional as F
from ..utils.cfgs_utils import get_value_from_cfgs_field, valid_key_in_cfgs

POTENTIAL_KEYS = ['img', 'mask', 'rays_o', 'rays_d', 'rays_r', 'bounds', 'bkg_color', 'exp_time']      # arcnerf/datasets/__init__.py:17
_CAMERA_KEYS = ('intrinsic', 'c2w')


def get_model_feed_in(inputs, device=None):
    """arcnerf/datasets/__init__.py:44-61: the model's keys of a batch (-> feed_in, batch_size); device 'gpu' moves host tensors to the
    GPU as the reference does, None leaves them where the dataset lives (a batch of this Pipeline is born on the dataset's device)"""
    feed_in = {}
    for key in POTENTIAL_KEYS:
        if key in inputs:
            feed_in[key] = inputs[key].cuda(non_blocking=True) if device == 'gpu' and not inputs[key].is_cuda else inputs[key]
    return feed_in, inputs['rays_o'].shape[0]


class _Log:
    def add_log(self, *a, **k):
        pass


class TrainView:
    """what `process_train_data` leaves in the dict under '_view': the dataset's tensors (untouched), the crop window and the permutation"""

    def __init__(self, tensors, n_img, H, W):
        self.tensors, self.n_img, self.H, self.W = tensors, n_img, H, W
        self.window = (0, 0, H, W)
        self.ids = None             # (total,) int64 device: row j of the reference's shuffled tensor = row ids[j] of its cropped tensor
        self.bad = None

    @property
    def per_img(self):
        return self.window[2] * self.window[3]


class Pipeline(object):
    def __init__(self, tape=None):
        """tape: feeds the random draws (tests): .shuffle(k, n, device) for the k-th randperm of step_ray_sample, .bkg(k, n_rays, device) for the
        k-th rand_like of fetch_step_bkg_color; default: torch's generator of the data's device"""
        self.train_sample_info = {'sample_mode': 'full', 'sample_cross_view': True, 'dynamic_batch_size': 0}
        self.crop_max_epoch = None
        self.init_precrop = False
        self.scheduler_cfg = None
        self.tape = tape
        self._n_shuffle, self._n_bkg = 0, 0

    def setup_cfgs(self, cfgs):
        self.scheduler_cfg = cfgs

    def set_info(self, key, value):
        self.train_sample_info[key] = value

    def get_info(self, key=None):
        return self.train_sample_info if key is None else self.train_sample_info[key]

    def set_n_rays(self, logger, n_rays):
        self.set_info('n_rays', n_rays)
        (logger or _Log()).add_log('Num of rays for each training batch: {}'.format(n_rays))

    def check_crop_shuffle(self, epoch):
        """pipeline.py:38-43"""
        return self.crop_max_epoch is not None and epoch >= self.crop_max_epoch

    def check_full_shuffle(self):
        """pipeline.py:45-52: every ray of a full-mode pass has been handed out"""
        return self.get_info('sample_mode') == 'full' and self.get_info('sample_total_count') >= self.get_info('total_samples')

    # ---- process_train_data ---------------------------------------------------------------------------------------------------------------
    def process_train_data(self, logger, train_data):
        """pipeline.py:54-93.  train_data: a dataset dict (see the module docstring) or the dict a previous call returned (the reshuffle of
        a finished pass, arcnerf_trainer.py:536-540).  Returns the dict with 'H' / 'W' = [0] and the view under '_view'."""
        logger = logger or _Log()
        self.check_bad_ids(train_data)
        self.set_info('sample_img_count', 0)
        self.set_info('sample_total_count', 0)
        train_data = self.step_crop_center_image(logger, train_data)
        train_data = self.step_ray_sample(logger, train_data)
        self.step_dynamic_bs(logger)
        self.step_bkg_color(logger, train_data)
        logger.add_log('Need {} epoch to run all the {} rays...'.format(
            math.ceil(float(self.get_info('total_samples')) / float(self.get_info('n_rays'))), self.get_info('total_samples')))
        train_data['H'] = [0]
        train_data['W'] = [0]
        return train_data

    @staticmethod
    def check_bad_ids(train_data):
        """arcn_fetch_train_batch clamps a ray id outside [0, n_img * window) to ray 0 and counts it (TrainView.bad): a wrong permutation or
        window would otherwise train on copies of pixel 0 without a word.  Read here - once per pass over the data (the reshuffle, the end of
        the crop), one host read - and raised."""
        view = train_data.get('_view') if isinstance(train_data, dict) else None
        if view is not None and view.bad is not None:
            n_bad = int(view.bad.item())
            if n_bad:
                raise RuntimeError('Pipeline: {} ray ids of the last pass were outside the dataset window (clamped to ray 0 by the batch fetch): '
                                   'the permutation or the crop window is wrong'.format(n_bad))

    @staticmethod
    def _view_of(train_data):
        """the view of a processed dict, or a fresh one over a dataset dict"""
        if '_view' in train_data:
            return train_data['_view'], False
        H, W = int(train_data['H']), int(train_data['W'])
        tensors = {k: v for k, v in train_data.items() if isinstance(v, torch.Tensor)}
        pix = [v for k, v in tensors.items() if k not in _CAMERA_KEYS]
        if not pix:
            raise RuntimeError('Pipeline: the dataset holds no per-pixel tensor')
        n_img = pix[0].shape[0]
        for k, v in tensors.items():
            if k in _CAMERA_KEYS:
                if v.shape[0] != n_img:
                    raise RuntimeError('Pipeline: {} must hold one camera per image'.format(k))
            elif v.shape[0] != n_img or v.shape[1] != H * W:
                raise RuntimeError('Pipeline: tensor {} is not (n_img, H*W, ...)'.format(k))
        return TrainView(tensors, n_img, H, W), True

    def step_crop_center_image(self, logger, train_data):
        """pipeline.py:95-130: the centre window of every image in the first call, when `scheduler.precrop.ratio` < 1; EVERY later call clears
        `crop_max_epoch` (the reference's state machine: its second call is either the end of the crop - the trainer hands in the dataset
        again - or the reshuffle of a finished pass, after which the crop never ends)"""
        view, fresh = self._view_of(train_data)
        if self.scheduler_cfg is not None and valid_key_in_cfgs(self.scheduler_cfg, 'precrop'):
            keep_ratio = get_value_from_cfgs_field(self.scheduler_cfg.precrop, 'ratio', 1.0)
            if keep_ratio < 1.0 and not self.init_precrop:
                self.init_precrop = True
                self.crop_max_epoch = get_value_from_cfgs_field(self.scheduler_cfg.precrop, 'max_epoch', None)
                if self.crop_max_epoch is not None:
                    logger.add_log('Crop sample on first {} epoch'.format(self.crop_max_epoch))
                logger.add_log('Crop training samples...keep ratio - {}'.format(keep_ratio))
                if not fresh:
                    raise RuntimeError('Pipeline: the first process_train_data call needs the dataset itself')
                h, w = view.H, view.W
                dh, dw = int((1 - keep_ratio) * h / 2.0), int((1 - keep_ratio) * w / 2.0)
                rows, cols = range(h)[dh:-dh], range(w)[dw:-dw]          # the reference's slice [dh:-dh, dw:-dw] (empty when dh == 0)
                if len(rows) == 0 or len(cols) == 0:
                    raise RuntimeError('Pipeline: precrop ratio {} leaves no pixel of a {} x {} image (the reference slices [{}:-{}])'.format(keep_ratio, h, w, dh, dh))
                view.window = (rows[0], cols[0], len(rows), len(cols))
            else:
                self.crop_max_epoch = None
        else:
            self.crop_max_epoch = None
        if fresh:
            # (a dataset handed in again - the end of the crop - is the whole image again: a fresh view has the full window)
            self.set_info('total_samples', view.n_img * view.per_img)
            self.set_info('n_train_img', view.n_img)
            self.set_info('n_train_hw', view.per_img)
            view.ids = None
        else:
            # the reference's tensors are (1, n_total, ...) from the first shuffle on
            if self.get_info('sample_mode') == 'full':
                self.set_info('n_train_img', 1)
                self.set_info('n_train_hw', self.get_info('total_samples'))
        train_data['_view'] = view
        return train_data

    def _randperm(self, n, device):
        if self.tape is not None:
            out = self.tape.shuffle(self._n_shuffle, n, device)
        else:
            out = torch.randperm(n, device=device)
        self._n_shuffle += 1
        return out

    def step_ray_sample(self, logger, train_data):
        """pipeline.py:132-174: `full` mode shuffles once per pass (cross view: one randperm over every ray; else batches of one image at a
        time); `random` mode draws at every batch"""
        if valid_key_in_cfgs(self.scheduler_cfg, 'ray_sample'):
            self.set_info('sample_mode', get_value_from_cfgs_field(self.scheduler_cfg.ray_sample, 'mode', 'full'))
            self.set_info('sample_cross_view', get_value_from_cfgs_field(self.scheduler_cfg.ray_sample, 'cross_view', True))
        assert self.get_info('sample_mode') in ['random', 'full'], 'Invalid mode {}'.format(self.get_info('sample_mode'))
        logger.add_log('Sample mode: {}, Cross view: {}'.format(self.get_info('sample_mode'), self.get_info('sample_cross_view')))
        view = train_data['_view']
        device = next(iter(view.tensors.values())).device
        if self.get_info('sample_mode') == 'full':
            if self.get_info('sample_cross_view'):
                random_idx = self._randperm(self.get_info('total_samples'), device)
            else:
                logger.add_log('Merge rays from different images into continuous batches..')
                n_train_hw, n_rays, n_train_img = self.get_info('n_train_hw'), self.get_info('n_rays'), self.get_info('n_train_img')
                per_img = self._randperm(n_train_hw, device)
                chunks = []
                for start in range(0, n_train_hw, n_rays):
                    for img_idx in self._randperm(n_train_img, device).tolist():
                        chunks.append(img_idx * n_train_hw + per_img[start:start + n_rays])
                random_idx = torch.cat(chunks, dim=0)
            # shuffling an already shuffled tensor composes the permutations
            view.ids = random_idx if view.ids is None else view.ids[random_idx]
        return train_data

    def step_dynamic_bs(self, logger=None):
        """pipeline.py:176-197: `dynamic_batch_size.update_epoch` / `max_batch_size` (default 32768) of the dataset scheduler block"""
        if valid_key_in_cfgs(self.scheduler_cfg, 'dynamic_batch_size') \
                and get_value_from_cfgs_field(self.scheduler_cfg.dynamic_batch_size, 'update_epoch', 0) > 0:
            self.set_info('dynamic_batch_size', self.scheduler_cfg.dynamic_batch_size.update_epoch)
            self.set_info('dynamic_max_batch_size', get_value_from_cfgs_field(self.scheduler_cfg.dynamic_batch_size, 'max_batch_size', 32768))
            assert not (self.get_info('sample_mode') == 'full' and not self.get_info('sample_cross_view')), 'Not allow full image without cross view'
        else:
            self.set_info('dynamic_batch_size', 0)

    def step_bkg_color(self, logger, train_data):
        if self._bkg_cfg(train_data) is not None:
            (logger or _Log()).add_log('Train with bkg color: {}'.format(self.scheduler_cfg.bkg_color.color))

    def _bkg_cfg(self, data):
        """the colour the scheduler blends in ('random' or [r, g, b]) when the data has a mask (pipeline.py:281-283), else None"""
        has_mask = 'mask' in data or 'rgba' in data or ('_view' in data and ('mask' in data['_view'].tensors or 'rgba' in data['_view'].tensors))
        if self.scheduler_cfg is not None and valid_key_in_cfgs(self.scheduler_cfg, 'bkg_color') and has_mask:
            return get_value_from_cfgs_field(self.scheduler_cfg.bkg_color, 'color', 'random')
        return None

    # ---- a batch --------------------------------------------------------------------------------------------------------------------------------
    def get_train_batch(self, train_data, epoch, model):
        """pipeline.py:204-221: dynamic batch size, the rays, the background colour, the non-tensor entries"""
        self.fetch_step_update_dynamic_bs(epoch, model)
        return self.fetch_batch(train_data)

    def fetch_batch(self, train_data):
        """get_train_batch without the batch-size update (trainer.train_epoch has done it): rays + colours + blend in one launch"""
        data_batch = self._fetch(train_data, {}, blend=True)
        return self.fetch_step_other_type(train_data, data_batch)

    def will_update_dynamic_bs(self, epoch):
        """the condition of fetch_step_update_dynamic_bs (pipeline.py:226-228) alone"""
        every = self.get_info('dynamic_batch_size')
        return every > 0 and epoch % every == 0 and epoch > 500

    def fetch_step_update_dynamic_bs(self, epoch, model):
        """pipeline.py:222-241: every `update_epoch` epochs AFTER epoch 500 the batch becomes n_rays x (the model's measured factor),
        rounded up to a multiple of 128 (on the float product, as the reference's `div_round_up` does) and capped.  `model`: anything with
        get_dynamicbs_factor() - a FullModel, a DDP-wrapped one (`.module`), or a trainer.DynamicBsMeter (`factor`)."""
        n_rays = self.get_info('n_rays')
        if self.get_info('dynamic_batch_size') > 0:
            update_epoch = self.get_info('dynamic_batch_size')
            if epoch % update_epoch == 0 and epoch > 500:
                if hasattr(model, 'get_dynamicbs_factor'):
                    dynamic_factor = model.get_dynamicbs_factor()
                elif hasattr(model, 'module'):
                    dynamic_factor = model.module.get_dynamicbs_factor()
                else:
                    dynamic_factor = model.factor()
                val = n_rays * dynamic_factor
                dynamic_n_rays = min(int((val + 128 - 1) // 128 * 128), self.get_info('dynamic_max_batch_size'))
                self.set_info('n_rays', dynamic_n_rays)
        return self.get_info('n_rays')

    def _batch_ids(self, view, device):
        """the rows of the reference's dataset tensor this batch holds (pipeline.py:243-277), as ids into the cropped tensor"""
        total_samples, n_rays = self.get_info('total_samples'), self.get_info('n_rays')
        if self.get_info('sample_mode') == 'random':
            if self.get_info('sample_cross_view'):
                return self._randperm(total_samples, device)[:n_rays]
            n_train_hw, n_train_img = self.get_info('n_train_hw'), self.get_info('n_train_img')
            img_idx = int(torch.randint(0, n_train_img, [1])[0])
            return img_idx * n_train_hw + self._randperm(n_train_hw, device)[:n_rays]
        count = self.get_info('sample_total_count')
        assert count < total_samples, 'All rays have been sampled, please reset train dataset...'
        ids = view.ids[count:count + n_rays]          # (the last batch of a pass is short, as the reference's slice is)
        self.set_info('sample_total_count', count + n_rays)
        return ids

    def _fetch(self, train_data, data_batch, blend):
        view = train_data['_view']
        t = view.tensors
        device = next(iter(t.values())).device
        ids = self._batch_ids(view, device)
        n = ids.shape[0]
        color = self._bkg_cfg(train_data) if blend else None
        bkg_rand = bkg_const = None
        if color == 'random':
            if self.tape is not None:
                bkg_rand = self.tape.bkg(self._n_bkg, n, device)
            else:
                bkg_rand = torch.rand((n, 3), dtype=torch.float32, device=device)
            self._n_bkg += 1
        elif color is not None:
            bkg_const = [float(c) for c in color]
        cams = 'intrinsic' in t and 'c2w' in t and 'rays_o' not in t
        rest = [k for k in t if k not in _CAMERA_KEYS + ('img', 'mask', 'rgba')]
        if view.bad is None:
            view.bad = torch.zeros(1, dtype=torch.int32, device=device)
        out = F.fetch_train_batch(ids, view.n_img, view.H, view.W, window=view.window, rgba=t.get('rgba'), img=t.get('img'), mask=t.get('mask'),
                                  intrinsic=t['intrinsic'] if cams else None, c2w=t['c2w'] if cams else None,
                                  center_pixel=bool(train_data.get('center_pixel', True)), normalize_rays_d=bool(train_data.get('normalize_rays_d', True)),
                                  bkg_rand=bkg_rand, bkg_const=bkg_const, want_src=bool(rest), bad_ids=view.bad)
        src = out.pop('src', None)
        for k, v in out.items():
            data_batch[k] = v.unsqueeze(0)
        for k in rest:                  # precomputed rays, bounds, exposure times ...: rows of the uncropped (n_img * H * W, ...) tensor
            v = t[k]
            data_batch[k] = v.reshape(-1, *v.shape[2:]).index_select(0, src).unsqueeze(0)
        return data_batch

    def fetch_step_ray_sample(self, train_data, data_batch):
        """pipeline.py:243-277 alone: the batch's rows without the background blend"""
        return self._fetch(train_data, data_batch, blend=False)

    def fetch_step_bkg_color(self, data_batch):
        """pipeline.py:279-300 alone, on a batch of fetch_step_ray_sample (get_train_batch does both in the one launch)"""
        color = self._bkg_cfg(data_batch)
        if color is None:
            return data_batch
        img, mask = data_batch['img'], data_batch['mask']
        if color == 'random':
            if self.tape is not None:
                bkg = self.tape.bkg(self._n_bkg, img.shape[-2], img.device).view_as(img)
            else:
                bkg = torch.rand_like(img)
            self._n_bkg += 1
        else:
            bkg = torch.ones_like(img) * torch.tensor(color, dtype=img.dtype, device=img.device)[None, None]
        data_batch['img'] = img * mask[..., None] + (1.0 - mask[..., None]) * bkg
        data_batch['bkg_color'] = bkg
        return data_batch

    @staticmethod
    def fetch_step_other_type(train_data, data_batch):
        """pipeline.py:302-309: the non-tensor entries ride along"""
        for k, v in train_data.items():
            if not isinstance(v, torch.Tensor) and not k.startswith('_'):
                data_batch[k] = v
        return data_batch


class TrainBatches:
    """`get_batch` of trainer.train_epoch on a Pipeline: the data side of the reference's train_epoch (arcnerf_trainer.py:531-546) - the end of
    the centre crop hands the dataset in again, a finished pass is reshuffled - then the batch and get_model_feed_in.  Batches must be
    asked for in epoch order, one per epoch (train_epoch draws the next ones early in that order)."""
    wants_epoch = True

    def __init__(self, pipeline, dataset_fn, logger=None):
        """dataset_fn() -> the dataset dict (ArcNerfTrainer.set_train_dataset; called again at the end of the crop: the tensors are never
        modified here, the same dict may be returned)"""
        self.pipeline, self.dataset_fn, self.logger = pipeline, dataset_fn, logger
        self.data = pipeline.process_train_data(logger, dict(dataset_fn()))
        self.drawn = []

    def __call__(self, n_rays, epoch):
        p = self.pipeline
        if p.check_crop_shuffle(epoch):
            self.data = p.process_train_data(self.logger, dict(self.dataset_fn()))
        elif p.check_full_shuffle():
            if not p.get_info('sample_cross_view'):
                self.data = dict(self.dataset_fn())
            self.data = p.process_train_data(self.logger, self.data)
        feed_in, _ = get_model_feed_in(p.fetch_batch(self.data))
        self.drawn.append((epoch, feed_in['rays_o'].shape[1]))
        return feed_in
