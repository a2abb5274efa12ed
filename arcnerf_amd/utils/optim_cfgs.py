"""The pruning / refresh settings (`epoch_optim`, `opa_thres`, ...) are read the same way by the object bounds
(basic_bound.py:30-62 in the reference) and by the background models (bkg_model.py:49-75): one reader, one accessor mixin."""
from .cfgs_utils import get_value_from_cfgs_field


def read_prune_settings(node, defaults):
    """{key: node.key if present else default}; `node` may be None"""
    return {k: get_value_from_cfgs_field(node, k, v) for k, v in defaults.items()}


class OptimCfgAccess:
    """expects `self.optim_cfgs` (a dict)"""

    def get_optim_cfgs(self, key=None):
        return self.optim_cfgs if key is None else self.optim_cfgs[key]

    def set_optim_cfgs(self, key, value):
        self.optim_cfgs[key] = value
