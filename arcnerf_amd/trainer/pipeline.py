"""Pipeline (arcnerf/trainer/pipeline.py:9-300), the part on the path: the number of rays of a training batch and its dynamic
adjustment.  (Ray shuffling / cropping of the dataset tensors is the data side, not mirrored.)"""
from ..utils.cfgs_utils import get_value_from_cfgs_field, valid_key_in_cfgs


class Pipeline(object):
    def __init__(self):
        self.train_sample_info = {'sample_mode': 'full', 'sample_cross_view': True, 'dynamic_batch_size': 0}
        self.scheduler_cfg = None

    def setup_cfgs(self, cfgs):
        self.scheduler_cfg = cfgs

    def set_info(self, key, value):
        self.train_sample_info[key] = value

    def get_info(self, key=None):
        return self.train_sample_info if key is None else self.train_sample_info[key]

    def set_n_rays(self, logger, n_rays):
        self.set_info('n_rays', n_rays)

    def step_dynamic_bs(self, logger=None):
        """pipeline.py:176-197: `dynamic_batch_size.update_epoch` / `max_batch_size` (default 32768) of the dataset scheduler block"""
        if valid_key_in_cfgs(self.scheduler_cfg, 'dynamic_batch_size') \
                and get_value_from_cfgs_field(self.scheduler_cfg.dynamic_batch_size, 'update_epoch', 0) > 0:
            self.set_info('dynamic_batch_size', self.scheduler_cfg.dynamic_batch_size.update_epoch)
            self.set_info('dynamic_max_batch_size', get_value_from_cfgs_field(self.scheduler_cfg.dynamic_batch_size, 'max_batch_size', 32768))
        else:
            self.set_info('dynamic_batch_size', 0)

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
